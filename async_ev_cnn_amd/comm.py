"""The multi-GPU side of the path: one process per GPU, independent streams, ONE collective -- the RCCL all-gather of the
detections (include/aev.h: aev_comm_*, aev_net_gather_detections).  No PyTorch: RCCL is bound by the library itself, so a rank
runs on the HIP runtime the library was built for.  The reference has no counterpart (it has no distributed code).

`rendezvous` is the side channel ncclUniqueId needs: rank 0 creates the id and serves it on a TCP socket, every other rank
fetches it.  Ranks, world size and address come from the launcher's environment (torch.distributed.run sets RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_ADDR / MASTER_PORT; the launcher's own store owns MASTER_PORT, so the id travels on MASTER_PORT + 1, or
AEV_RDV_PORT)."""
import ctypes as C
import os
import socket
import struct
import time

import numpy as np

from . import _native as N

ID_BYTES = 128
SUM, MAX = 0, 1


def rendezvous_port():
    return int(os.environ.get("AEV_RDV_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))


PORT_SPAN = 16   # rank 0 serves on the first free port of [port, port + PORT_SPAN); the others probe the same range


def exchange_id(rank, world, make_id, addr=None, port=None, timeout=300.0):
    """Rank 0: `make_id()` -> bytes, served to the world - 1 other ranks; other ranks: fetch it.  Returns the id (bytes).
    Every client sends a magic word and its rank first and gets the magic word back in front of the id, so that rank 0 can tell a
    stray connection from a peer (and knows when all have been served) and a client can tell rank 0 from whatever else may listen
    on a port of the range."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = rendezvous_port() if port is None else port
    if rank == 0:
        uid = bytes(make_id())
        if world == 1:
            return uid
        srv, err = None, None
        for p in range(port, port + PORT_SPAN):
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind(("0.0.0.0" if addr not in ("127.0.0.1", "localhost") else "127.0.0.1", p))
                break
            except OSError as e:
                srv.close()
                srv, err = None, e
        if srv is None:
            raise RuntimeError("rank 0: no free rendezvous port in %d..%d (%s)" % (port, port + PORT_SPAN - 1, err))
        srv.listen(world)
        srv.settimeout(timeout)
        served = set()
        try:
            while len(served) < world - 1:
                conn, _ = srv.accept()
                with conn:
                    try:
                        conn.settimeout(10.0)
                        magic, peer = struct.unpack("<4si", _recv_exact(conn, 8))
                        if magic != b"AEV1" or not 0 < peer < world:
                            continue
                        conn.sendall(struct.pack("<4si", b"AEV1", len(uid)) + uid)
                        served.add(peer)
                    except (OSError, struct.error):
                        continue            # (a stray or broken connection: the peer will come again)
        finally:
            srv.close()
        return uid
    deadline = time.time() + timeout
    while True:
        for p in range(port, port + PORT_SPAN):
            try:
                with socket.create_connection((addr, p), timeout=2.0) as conn:
                    conn.settimeout(10.0)
                    conn.sendall(struct.pack("<4si", b"AEV1", rank))
                    magic, n = struct.unpack("<4si", _recv_exact(conn, 8))
                    if magic != b"AEV1" or n != ID_BYTES:
                        continue
                    return _recv_exact(conn, n)
            except (OSError, struct.error):
                continue
        if time.time() > deadline:
            raise RuntimeError("rank %d: no rendezvous with rank 0 at %s:%d..%d within %.0f s" % (rank, addr, port, port + PORT_SPAN - 1, timeout))
        time.sleep(0.05)


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        part = conn.recv(n - len(buf))
        if not part:
            raise ConnectionResetError("peer closed the rendezvous socket")
        buf += part
    return buf


class Comm:
    """An RCCL communicator over the ranks of one job (aev_comm_create).  `Comm.from_env()` does the rendezvous."""

    def __init__(self, rank, world, uid, device):
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self.handle = C.c_void_p()
        buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid)
        N.check(N.lib().aev_comm_create(self.rank, self.world, buf, self.device, C.byref(self.handle)), "aev_comm_create")

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * ID_BYTES)()
        N.check(N.lib().aev_comm_unique_id(buf), "aev_comm_unique_id")
        return bytes(buf)

    @classmethod
    def from_env(cls, device=None):
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        uid = exchange_id(rank, world, cls.unique_id)
        return cls(rank, world, uid, N.default_device() if device is None else device)

    def close(self):
        if getattr(self, "handle", None):
            N.lib().aev_comm_destroy(self.handle)
            self.handle = None

    __del__ = close

    def allreduce(self, values, op=SUM):
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        N.check(N.lib().aev_comm_allreduce(self.handle, v.ctypes.data_as(C.POINTER(C.c_double)), len(v), int(op)), "aev_comm_allreduce")
        return v

    def barrier(self):
        self.allreduce([0.0])

    def gather_detections(self, net, num_classes, num_bbox, conf_threshold=0.1, use_nms=True, nms_threshold=0.0, s_cap=None):
        """([world, S_max, nb, 7] float32, counts): every rank's detections (EventNet.detect), rows behind a rank's count are zero."""
        c, h, w = net.shapes[-1]
        s_cap = int(s_cap or net.n_streams + 1)     # (stream i -> rank i mod world: counts differ by one at most)
        out = np.zeros((self.world, s_cap, h * w * num_bbox, 7), np.float32)
        counts = np.zeros(self.world, np.int32)
        smax = C.c_int(0)
        N.check(N.lib().aev_net_gather_detections(net.handle, self.handle, int(num_classes), int(num_bbox), float(conf_threshold),
                                                  1 if use_nms else 0, float(nms_threshold), s_cap, out.ctypes.data_as(N.f32p),
                                                  counts.ctypes.data_as(N.i32p), C.byref(smax)), "aev_net_gather_detections")
        flat = out.reshape(-1)[:self.world * smax.value * h * w * num_bbox * 7]
        return flat.reshape(self.world, smax.value, h * w * num_bbox, 7), [int(v) for v in counts]
