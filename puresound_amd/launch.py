"""One process per GPU from a single command (replaces the one-call multi-GPU of the reference,
`nn.DataParallel` in puresound/task/base.py:226-229).

`python script.py --gpus N` without an outer launcher: the parent -- which must not have touched the GPU, so this runs
before anything initialises HIP -- starts `python -m torch.distributed.run --nproc-per-node N script.py <same arguments>`
as a CHILD process (never an exec: a process image must not be replaced once a GPU may have been initialised), relays its
output and returns its exit status.  Under an outer `torch.distributed.run` (WORLD_SIZE set) nothing happens here.
"""
import os
import socket
import subprocess
import sys


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def needs_self_launch(gpus, env=None):
    env = os.environ if env is None else env
    return gpus > 1 and "WORLD_SIZE" not in env


def self_launch(script, argv, gpus, env=None):
    """Run `script argv...` as `gpus` ranks on this node; returns the launcher's exit status.  stdout / stderr are the
    children's (rank 0 prints the result line)."""
    env = dict(os.environ if env is None else env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    return subprocess.run(cmd, env=env).returncode


def launch_probe(tag):
    """Test hook (PS_LAUNCH_PROBE=1): a rank reports what the launcher gave it and leaves before touching a GPU."""
    if os.environ.get("PS_LAUNCH_PROBE"):
        import json
        print(json.dumps({"probe": tag, "world_size": int(os.environ.get("WORLD_SIZE", "1")),
                          "rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                          "master_addr": os.environ.get("MASTER_ADDR")}), flush=True)
        return True
    return False
