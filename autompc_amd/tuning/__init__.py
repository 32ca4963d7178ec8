from .batch_eval import (CandidateEvaluator, evaluate_sharded, random_candidates,
                         score_trajectories, shard_bounds)
from .batch_tuner import BatchPipelineTuner, PipelineTuneResult

__all__ = ["CandidateEvaluator", "evaluate_sharded", "random_candidates", "score_trajectories",
           "shard_bounds", "BatchPipelineTuner", "PipelineTuneResult"]
