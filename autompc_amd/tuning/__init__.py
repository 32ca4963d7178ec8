from .batch_eval import (CandidateEvaluator, IlqrCandidateEvaluator, balanced_shards, candidate_work,
                         evaluate_sharded, random_candidates, random_ilqr_candidates, score_trajectories,
                         shard_bounds)
from .batch_tuner import BatchPipelineTuner, PipelineTuneResult

__all__ = ["CandidateEvaluator", "IlqrCandidateEvaluator", "balanced_shards", "candidate_work", "evaluate_sharded", "random_candidates",
           "random_ilqr_candidates", "score_trajectories", "shard_bounds", "BatchPipelineTuner",
           "PipelineTuneResult"]
