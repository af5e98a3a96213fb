"""Child process of tests/test_gpu_sharded.py: sharding.sharded_align with the HIP engine over a REAL RCCL process group
(backend "nccl", world size 1: one GPU on the test box), compared with the plain align() of the same engine.

The process group is initialised before anything else touches the GPU.  No manual synchronisation anywhere: the pass kernels,
the reduce-only solver, the all-reduce and the solver step are ordered by the one stream sharded_align runs them on.
usage: python tests/_sharded_rccl_worker.py <port> <n_points>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port, n = int(sys.argv[1]), int(sys.argv[2])
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    torch.cuda.set_device(0)
    import numpy as np
    from direct_lidar_odometry_amd import clouds, sharding
    from direct_lidar_odometry_amd.nano_gicp import NanoGICP

    w = clouds.scan_to_scan(n)
    full = NanoGICP(); full.setMaxCorrespondenceDistance(1.0); full.setInputSource(w.source); full.setInputTarget(w.target)
    full.align()
    cs, ct = full.getSourceCovariances(), full.getTargetCovariances()
    out = {"runs": []}
    for rep in range(3):  # repeated: a stream-ordering race would not fail the same way every time
        lo, hi = sharding.shard_bounds(len(w.source), dist.get_world_size(), dist.get_rank())
        e = NanoGICP(); e.setMaxCorrespondenceDistance(1.0)
        e.setInputSource(w.source[lo:hi]); e.setInputTarget(w.target)
        e.setSourceCovariances(cs[lo:hi]); e.setTargetCovariances(ct)
        # something unrelated in flight on torch's default stream while the loop runs on its own stream
        junk = torch.randn(1 << 22, device="cuda:0")
        for _ in range(4):
            junk = junk * 1.0001 + 0.5
        T = sharding.sharded_align(e, None, dist, "cuda:0")
        out["runs"].append({"bit_equal": bool(np.array_equal(T, full.getFinalTransformation())),
                            "iters": [int(e.nr_iterations_), int(full.nr_iterations_)],
                            "converged": [bool(e.converged_), bool(full.converged_)],
                            "max_abs_diff": float(np.abs(T - full.getFinalTransformation()).max())})
        e.close()
    # K1 sharded (world size 1: the whole array is this rank's block; the tensor aliases the engine's device buffer)
    a = NanoGICP(); a.setCorrespondenceRandomness(10); a.setInputSource(w.source)
    n_cov = sharding.sharded_covariances(a, 0, dist, "cuda:0")
    b = NanoGICP(); b.setCorrespondenceRandomness(10); b.setInputSource(w.source); b.calculateSourceCovariances()
    out["covs"] = {"n": int(n_cov), "bit_equal": bool(np.array_equal(a.getSourceCovariances(), b.getSourceCovariances()))}
    ptr, nn = a.covsShardBegin(0)
    view = sharding.device_doubles(ptr, nn * 6, "cuda:0")
    out["covs"]["alias_len"] = int(view.numel())
    out["backend"] = dist.get_backend()
    out["world_size"] = dist.get_world_size()
    dist.destroy_process_group()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
