from .batch_eval import (CandidateEvaluator, evaluate_sharded, random_candidates,
                         score_trajectories, shard_bounds)

__all__ = ["CandidateEvaluator", "evaluate_sharded", "random_candidates", "score_trajectories",
           "shard_bounds"]
