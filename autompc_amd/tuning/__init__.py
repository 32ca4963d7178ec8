from .batch_eval import (CandidateEvaluator, IlqrCandidateEvaluator, balanced_shards, candidate_work,
                         evaluate_sharded, random_candidates, random_ilqr_candidates, score_trajectories,
                         shard_bounds)
from .batch_tuner import BatchPipelineTuner, PipelineTuneResult
from .configs import (DictConfiguration, candidate_from_config, candidates_from_configs, config_from_candidate,
                      sample_pipeline_configs)

__all__ = ["CandidateEvaluator", "IlqrCandidateEvaluator", "balanced_shards", "candidate_work", "evaluate_sharded", "random_candidates",
           "random_ilqr_candidates", "score_trajectories", "shard_bounds", "BatchPipelineTuner",
           "PipelineTuneResult", "DictConfiguration", "candidate_from_config", "candidates_from_configs",
           "config_from_candidate", "sample_pipeline_configs"]
