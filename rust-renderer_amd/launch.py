"""One process per GPU: start N ranks of a script (spawn_ranks: under torch.distributed.run, the launcher the benchmark driver
uses, rendezvous on 127.0.0.1) and let them find each other WITHOUT torch (Rendezvous below).

The caller must not have touched the GPU: the ranks are fresh child processes (subprocess, never an exec of the running
program), each of which binds its own device. The parent only waits and hands the children's exit status on; rank 0's stdout
is the job's stdout.

A rank is a GPU process and imports no torch: the one thing the ranks must exchange before RCCL exists is the 128-byte
ncclUniqueId rank 0 makes (uh_rccl_unique_id -> uh_rccl_attach), plus a barrier and a few numbers around the benchmark's
timed region. Rendezvous does that over plain TCP sockets on 127.0.0.1 (one node: the contract of bench.py). The launcher's
own store listens on MASTER_PORT, so rank 0 listens on an ephemeral port and publishes it in a file named after the job
(MASTER_PORT + the launcher's run id) in the temp directory.
"""
import os
import socket
import struct
import subprocess
import sys
import tempfile
import time


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


_MAGIC = b"UHRDZV1\0"


def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


def _pack_list(blobs):
    return struct.pack("<I", len(blobs)) + b"".join(struct.pack("<Q", len(b)) + b for b in blobs)


def _unpack_list(data):
    (n,), at, out = struct.unpack_from("<I", data, 0), 4, []
    for _ in range(n):
        (k,) = struct.unpack_from("<Q", data, at)
        out.append(data[at + 8:at + 8 + k])
        at += 8 + k
    return out


class Rendezvous:
    """The ranks of ONE node find each other over TCP on 127.0.0.1: rank 0 listens on an ephemeral port and publishes it in
    <tmp>/utopian_rdzv_<key>; every other rank connects and says who it is. Star topology through rank 0; every operation is
    collective (all ranks call it, in the same order). `key` names the job: from_env() derives it from the launcher's
    environment (MASTER_PORT, TORCHELASTIC_RUN_ID, TORCHELASTIC_RESTART_COUNT)."""

    def __init__(self, rank, world, key, timeout=180.0):
        self.rank, self.world, self.key = int(rank), int(world), str(key)
        self.peers = {}   # rank 0: rank -> socket
        self.root = None  # other ranks: socket to rank 0
        self.path = os.path.join(tempfile.gettempdir(), f"utopian_rdzv_{self.key}")
        if self.world <= 1:
            return
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            try:
                os.unlink(self.path)  # a stale file of an earlier job with the same key
            except FileNotFoundError:
                pass
            srv = socket.socket()
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            tmp = f"{self.path}.{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(str(srv.getsockname()[1]))
            os.replace(tmp, self.path)
            srv.settimeout(1.0)
            while len(self.peers) < self.world - 1:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous {self.key}: {len(self.peers) + 1} of {self.world} ranks arrived")
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                try:
                    conn.settimeout(5.0)
                    msg = _recv(conn)
                    r = struct.unpack_from("<I", msg, len(_MAGIC) + 4)[0]
                    if msg[:len(_MAGIC)] != _MAGIC or msg[len(_MAGIC) + 8:] != self.key.encode() or not (0 < r < self.world) or r in self.peers:
                        raise ValueError("not a rank of this job")
                    _send(conn, _MAGIC)
                    conn.settimeout(None)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.peers[r] = conn
                except Exception:
                    conn.close()
            srv.close()
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
        else:
            hello = _MAGIC + struct.pack("<II", self.world, self.rank) + self.key.encode()
            while self.root is None:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous {self.key}: rank {self.rank} found no rank 0")
                try:
                    port = int(open(self.path).read().strip())
                    s = socket.create_connection(("127.0.0.1", port), timeout=2.0)
                    try:
                        s.settimeout(5.0)
                        _send(s, hello)
                        if _recv(s) != _MAGIC:
                            raise ValueError("not this job's rank 0")
                        s.settimeout(None)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self.root = s
                    except Exception:
                        s.close()
                        raise
                except Exception:
                    time.sleep(0.05)  # no file yet, a stale file, or rank 0 not listening yet

    @classmethod
    def from_env(cls, environ=None, timeout=180.0):
        e = os.environ if environ is None else environ
        key = "_".join(str(e.get(k, "none")) for k in ("MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"))
        return cls(int(e.get("RANK", "0")), int(e.get("WORLD_SIZE", "1")), key, timeout)

    def allgather(self, data):
        """every rank's bytes, in rank order, on every rank"""
        data = bytes(data)
        if self.world <= 1:
            return [data]
        if self.rank == 0:
            blobs = [data] + [_recv(self.peers[r]) for r in range(1, self.world)]
            packed = _pack_list(blobs)
            for r in range(1, self.world):
                _send(self.peers[r], packed)
            return blobs
        _send(self.root, data)
        return _unpack_list(_recv(self.root))

    def broadcast(self, data, src=0):
        """rank src's bytes on every rank (the others pass None)"""
        return self.allgather(data if self.rank == src else b"")[src]

    def gather(self, data, dst=0):
        """every rank's bytes on rank dst (None elsewhere)"""
        blobs = self.allgather(data)
        return blobs if self.rank == dst else None

    def barrier(self):
        self.allgather(b"")

    def allreduce(self, values, op="sum"):
        """element-wise sum / max / min of a list of floats over the ranks (exact doubles: they travel as 8 bytes each)"""
        values = [float(v) for v in values]
        rows = [struct.unpack(f"<{len(values)}d", b) for b in self.allgather(struct.pack(f"<{len(values)}d", *values))]
        f = {"sum": sum, "max": max, "min": min}[op]
        return [f(col) for col in zip(*rows)]

    def close(self):
        for s in list(self.peers.values()) + ([self.root] if self.root else []):
            try:
                s.close()
            except Exception:
                pass
        self.peers, self.root = {}, None
