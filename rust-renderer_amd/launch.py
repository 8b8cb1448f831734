"""One process per GPU: start N ranks of a script under torch.distributed.run (rendezvous on 127.0.0.1).

The caller must not have touched the GPU: the ranks are fresh child processes (subprocess, never an
exec of the running program), each of which binds its own device. The parent only waits and hands the
children's exit status on; rank 0's stdout is the job's stdout.
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def under_launcher(environ=None):
    """True inside a rank started by torch.distributed.run / torchrun (RANK and WORLD_SIZE are set)."""
    environ = os.environ if environ is None else environ
    return "RANK" in environ and "WORLD_SIZE" in environ


def rank_command(nproc, script, argv, port=None):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), script] + list(argv)


def spawn_ranks(nproc, script, argv, port=None, env=None, timeout=None, **popen_kw):
    """Run `script argv` as nproc ranks; returns the launcher's exit status (0 only if every rank exited 0)."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    return subprocess.run(rank_command(nproc, script, argv, port), env=e, timeout=timeout, **popen_kw).returncode
