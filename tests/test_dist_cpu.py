"""world_size-2 test of the N>1 path on CPU (gloo): each rank takes its shard from the
partitioner, computes shard loss/gradients normalised by the GLOBAL batch (here with the CPU
oracle as the compute -- the HIP path needs a GPU), all-reduces the flat gradient buffer with
the two metric floats riding at its tail, applies SGD, and must land on the single-rank result.
This is the exact host sequence GCN2.train_step runs with RCCL on the GPU box."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "gcn-string_amd"))
    import torch, torch.distributed as dist
    from oracle import gcn_oracle as O
    from gcnx import shard, synth
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
    rank = dist.get_rank()
    hb = synth.ecoli_batch(5, 8, seed=21)
    hb.vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    params = O.gcn2_init(np.random.default_rng(0), 8, 8, 2)
    s, gb = shard.shard_batch(hb, rank, 2)
    f64 = lambda b: (b.x.astype(np.float64), (b.rowptr.astype(np.int64), b.colidx.astype(np.int64), b.vals.astype(np.float64)), b.graph_ptr, b.y.astype(np.float64))
    x, csr, gp, y = f64(s)
    loss, acc, g, _ = O.gcn2_loss_and_grads(params, x, csr, gp, y, denom=gb)
    flat = np.concatenate([g[k].ravel() for k in O.GCN2_PARAM_ORDER] + [[loss, acc * s.n_graphs]])
    t = torch.from_numpy(flat); dist.all_reduce(t)           # one fused all-reduce (SURVEY 8(e))
    flat = t.numpy()
    pflat = np.concatenate([params[k].ravel() for k in O.GCN2_PARAM_ORDER]) - 0.02 * flat[:-2]
    if rank == 0:
        x, csr, gp, y = f64(hb)
        l1, a1, g1, _ = O.gcn2_loss_and_grads(params, x, csr, gp, y)
        ref = np.concatenate([params[k].ravel() - 0.02 * g1[k].ravel() for k in O.GCN2_PARAM_ORDER])
        assert abs(flat[-2] - l1) < 1e-12 and abs(flat[-1] / gb - a1) < 1e-12
        assert np.max(np.abs(pflat - ref)) < 1e-13
        print("DIST_OK")
    dist.barrier(); dist.destroy_process_group()
""")


def test_two_rank_gloo_step_matches_single_rank(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, port=port))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "DIST_OK" in outs[0]
