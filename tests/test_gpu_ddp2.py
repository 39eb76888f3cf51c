"""GPU: the real ddp.Trainer with two ranks (one process per rank, started by torch.distributed.run before either
touches the GPU; both ranks share the box's one GPU, collectives over gloo): tools/ddp2_check.py holds the checks --
per-shard logits against the oracle, the averaged gradient against the mean of the per-shard oracle gradients,
buckets launched from the backward hooks, replicas bit-identical after the update, the HIP-graph path."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trainer_two_ranks(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, STGCN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", STGCN_DDP2_VERBOSE="1", STGCN_DDP2_OUT=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "ddp2_check.py")]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=170, text=True)
    print("\n".join(l for l in p.stdout.splitlines() if l.startswith("[rank") or "DDP2" in l or "launch-by" in l or "graph replay" in l))
    assert p.returncode == 0 and "DDP2_RANKS_OK" in p.stdout, p.stdout[-3000:]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ddp2_check
    ddp2_check.verify(str(tmp_path))


def test_trainer_over_rccl_one_rank():
    """The RCCL code path itself on the box's one GPU: a one-rank "nccl" group (tools/rccl_one_rank.py)."""
    env = dict(os.environ, STGCN_DDP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")

    def launch():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tools", "rccl_one_rank.py")]
        return subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=170, text=True)

    p = launch()
    port_taken = any(t in p.stdout for t in ("Address already in use", "address already in use", "EADDRINUSE",
                                             "failed to listen on", "RendezvousConnectionError"))
    if p.returncode != 0 and "RCCL_ONE_RANK_GROUP_UP" not in p.stdout and port_taken:
        # the launcher could not bind its rendezvous port (the probed port can be taken between the probe and the
        # bind): one more launch on a fresh port.  Anything else -- init_process_group("nccl") failing, a failure
        # after the group is up -- is the code's own and fails the test as it is.
        print("rendezvous port taken, launching once more:\n" + p.stdout[-1500:])
        p = launch()
    print("\n".join(l for l in p.stdout.splitlines() if "RCCL" in l or "graph=" in l))
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in p.stdout, p.stdout[-3000:]
