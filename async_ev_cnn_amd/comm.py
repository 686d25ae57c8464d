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


# ------------------------------------------------------------------------------------------------ the host-side control plane
class TcpComm:
    """A star of TCP connections through rank 0 with the interface of `Comm` (allreduce / barrier / gather_detections): the job's
    control plane while the RCCL communicator is being set up -- the ncclUniqueId travels over it, and the ranks agree over it
    whether every one of them has a communicator -- and the fallback transport of the final gather if RCCL cannot be brought up
    (the bench line then says so: a job that has measured N GPUs should not lose its numbers to the transport of 2 KB per stream).
    Never on the compute path: the streams are independent.  Rank 0 listens on the first free port of
    [rendezvous_port() + PORT_SPAN, + 2 PORT_SPAN); peers probe the range with a magic-word handshake (as exchange_id does)."""
    MAGIC = b"AEV2"

    def __init__(self, rank, world, addr=None, port=None, timeout=900.0):   # (ranks may reach their first collective minutes apart: each generates its streams first)
        self.rank, self.world = int(rank), int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = rendezvous_port() + PORT_SPAN if port is None else port
        self.socks = {}
        if self.world == 1:
            return
        if self.rank == 0:
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
                raise RuntimeError("rank 0: no free control-plane port in %d..%d (%s)" % (port, port + PORT_SPAN - 1, err))
            srv.listen(self.world)
            srv.settimeout(timeout)
            try:
                while len(self.socks) < self.world - 1:
                    conn, _ = srv.accept()
                    try:
                        conn.settimeout(10.0)
                        magic, peer = struct.unpack("<4si", _recv_exact(conn, 8))
                        if magic != self.MAGIC or not 0 < peer < self.world or peer in self.socks:
                            conn.close()
                            continue
                        conn.sendall(struct.pack("<4si", self.MAGIC, self.world))
                        conn.settimeout(timeout)
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self.socks[peer] = conn
                    except (OSError, struct.error):
                        conn.close()
            finally:
                srv.close()
            return
        deadline = time.time() + timeout
        while not self.socks:
            for p in range(port, port + PORT_SPAN):
                conn = None
                try:
                    conn = socket.create_connection((addr, p), timeout=2.0)
                    conn.settimeout(10.0)
                    conn.sendall(struct.pack("<4si", self.MAGIC, self.rank))
                    magic, w = struct.unpack("<4si", _recv_exact(conn, 8))
                    if magic == self.MAGIC and w == self.world:
                        conn.settimeout(timeout)
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self.socks[0] = conn
                        break
                    conn.close()
                except (OSError, struct.error):
                    if conn is not None:
                        conn.close()
            if not self.socks:
                if time.time() > deadline:
                    raise RuntimeError("rank %d: no control-plane connection to rank 0 at %s:%d..%d within %.0f s"
                                       % (self.rank, addr, port, port + PORT_SPAN - 1, timeout))
                time.sleep(0.05)

    @classmethod
    def from_env(cls, **kw):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), **kw)

    @staticmethod
    def _send(conn, payload):
        conn.sendall(struct.pack("<q", len(payload)) + payload)

    @staticmethod
    def _recv(conn):
        return _recv_exact(conn, struct.unpack("<q", _recv_exact(conn, 8))[0])

    def allgather_bytes(self, payload):
        """every rank's bytes, in rank order, on every rank"""
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [self._recv(self.socks[r]) for r in range(1, self.world)]
            blob = b"".join(struct.pack("<q", len(p)) + p for p in parts)
            for r in range(1, self.world):
                self._send(self.socks[r], blob)
            return parts
        self._send(self.socks[0], payload)
        blob, parts, o = self._recv(self.socks[0]), [], 0
        for _ in range(self.world):
            n = struct.unpack_from("<q", blob, o)[0]
            parts.append(blob[o + 8:o + 8 + n])
            o += 8 + n
        return parts

    def bcast_bytes(self, payload=b""):
        """rank 0's bytes on every rank"""
        return self.allgather_bytes(payload if self.rank == 0 else b"")[0]

    def allreduce(self, values, op=SUM):
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64)
        parts = np.stack([np.frombuffer(p, np.float64) for p in self.allgather_bytes(v.tobytes())])
        return parts.max(0) if op == MAX else parts.sum(0)

    def barrier(self):
        self.allreduce([0.0])

    def gather_detections(self, net, num_classes, num_bbox, conf_threshold=0.1, use_nms=True, nms_threshold=0.0, s_cap=None):
        """As Comm.gather_detections: ([world, S_max, nb, 7] float32, counts); the detections are decoded on the rank's GPU
        (EventNet.detect) and travel as host bytes."""
        d = np.ascontiguousarray(net.detect(num_classes, num_bbox, conf_threshold, use_nms, nms_threshold), dtype=np.float32)
        parts = [np.frombuffer(p, np.float32).reshape((-1,) + d.shape[1:]) for p in self.allgather_bytes(d.tobytes())]
        counts = [int(p.shape[0]) for p in parts]
        out = np.zeros((self.world, max(counts)) + d.shape[1:], np.float32)
        for r, p in enumerate(parts):
            out[r, :counts[r]] = p
        return out, counts

    def close(self):
        for c in self.socks.values():
            try:
                c.close()
            except OSError:
                pass
        self.socks = {}


def open_comm(device=None, rccl_timeout=240.0, log=None):
    """The communicator of this job's ranks -> (comm, backend): an RCCL communicator (`Comm`, backend "rccl") if EVERY rank gets one
    within `rccl_timeout` seconds, else the TCP star (`TcpComm`, backend "tcp: <why>") on every rank.  The ranks first connect the
    TCP star (plain sockets: nothing GPU-side happens before the launcher's processes have found each other), rank 0's ncclUniqueId
    travels over it, ncclCommInitRank runs in a thread of its own (a rank that cannot reach its peers blocks in it), and the ranks
    agree on the outcome over the star."""
    import threading
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = N.default_device() if device is None else device
    tcp = TcpComm(rank, world)
    uid, why = b"", ""
    if rank == 0:
        try:
            uid = Comm.unique_id()
        except Exception as e:      # (librccl not found, no device ...)
            why = "ncclGetUniqueId: %s" % e
    uid = tcp.bcast_bytes(uid)
    box = {}
    if len(uid) == ID_BYTES:
        def create():
            try:
                box["comm"] = Comm(rank, world, uid, device)
            except Exception as e:
                box["err"] = "rank %d: %s" % (rank, e)
        th = threading.Thread(target=create, daemon=True)
        th.start()
        th.join(rccl_timeout)
        if th.is_alive():
            box["err"] = "rank %d: ncclCommInitRank did not return within %.0f s" % (rank, rccl_timeout)
    else:
        box["err"] = "rank 0: no ncclUniqueId"
    errs = [p.decode("utf-8", "replace") for p in tcp.allgather_bytes(box.get("err", "").encode()) if p]
    if not errs and "comm" in box:
        tcp.close()
        return box["comm"], "rccl"
    why = why or errs[0]
    if log:
        log("[aev comm] no RCCL communicator (%s): the control plane and the final gather use the TCP star" % why)
    if "comm" in box:
        box["comm"].close()
    return tcp, "tcp: " + why
