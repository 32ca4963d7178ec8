"""The tuner's model axis: fit_s against eval_s for 64 MPPI candidates over 8 distinct 2x256 MLPs
fitted on one trajectory set -- bench.py's `sub_records.model_axis`, stand-alone, with the knobs exposed.
python tools/model_axis_rate.py [n_models] [epochs] [n_traj]"""
import argparse
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench                                                          # noqa: E402


class _R:
    local_rank, rank, world = 0, 0, 1

    @staticmethod
    def sync_all():
        import torch
        torch.cuda.synchronize()


n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 8
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n_traj = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rec = bench.model_axis_record(argparse.Namespace(), _R, n_models=n_models, epochs=epochs, n_traj=n_traj)
print(json.dumps(rec, indent=1))
