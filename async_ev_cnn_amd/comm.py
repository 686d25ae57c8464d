"""The multi-GPU side of the path: one process per GPU, independent streams, ONE collective -- the RCCL all-gather of the
detections (include/aev.h: aev_comm_*, aev_net_gather_detections).  No PyTorch: RCCL is bound by the library itself, so a rank
runs on the HIP runtime the library was built for.  The reference has no counterpart (it has no distributed code).

`rendezvous` is the side channel ncclUniqueId needs: rank 0 creates the id and serves it on a TCP socket, every other rank
fetches it.  Ranks, world size and address come from the launcher's environment (torch.distributed.run sets RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_ADDR / MASTER_PORT; the launcher's own store owns MASTER_PORT, so the id travels on MASTER_PORT + 1, or
AEV_RDV_PORT)."""
import ctypes as C
import hashlib
import os
import select
import socket
import struct
import subprocess
import sys
import time

import numpy as np

from . import _native as N

ID_BYTES = 128
SUM, MAX = 0, 1


def rendezvous_port():
    return int(os.environ.get("AEV_RDV_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))


PORT_SPAN = 16   # rank 0 serves on the first free port of [port, port + PORT_SPAN); the others probe the same range


def job_token():
    """8 bytes that identify THIS job in both handshakes: two jobs on one node whose MASTER_PORTs lie within 2 * PORT_SPAN of each other
    have overlapping port windows, and a magic word alone would let a client of one join the rank 0 of the other (same world size:
    ranks of different jobs would reduce and gather together).  MASTER_PORT is unique per job on a node (the launcher's store binds
    it); the launcher's run id and AEV_JOB_TOKEN (any string, for launchers without one) tell jobs on different nodes apart."""
    key = "|".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "AEV_JOB_TOKEN"))
    return hashlib.sha256(key.encode()).digest()[:8]


HS = "<4s8si"    # handshake: magic, job token, rank (client -> rank 0) / payload size or world (rank 0 -> client)
HS_BYTES = struct.calcsize(HS)


def exchange_id(rank, world, make_id, addr=None, port=None, timeout=300.0):
    """Rank 0: `make_id()` -> bytes, served to the world - 1 other ranks; other ranks: fetch it.  Returns the id (bytes).
    Every client sends a magic word, the job's token (job_token) and its rank first and gets magic word and token back in front of
    the id, so that rank 0 can tell a stray connection -- or a rank of ANOTHER job probing an overlapping port window -- from a peer
    (and knows when all have been served) and a client can tell its own rank 0 from whatever else may listen on a port of the range
    (a mismatch is "not my rank 0": it keeps probing)."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = rendezvous_port() if port is None else port
    token = job_token()
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
                        magic, tok, peer = struct.unpack(HS, _recv_exact(conn, HS_BYTES))
                        if magic != b"AEV1" or tok != token or not 0 < peer < world:
                            continue
                        conn.sendall(struct.pack(HS, b"AEV1", token, len(uid)) + uid)
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
                    conn.sendall(struct.pack(HS, b"AEV1", token, rank))
                    magic, tok, n = struct.unpack(HS, _recv_exact(conn, HS_BYTES))
                    if magic != b"AEV1" or tok != token or n != ID_BYTES:
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
    [rendezvous_port() + PORT_SPAN, + 2 PORT_SPAN); peers probe the range with a magic-word + job-token handshake (as exchange_id
    does: a rank of another job on the node, probing an overlapping window, is turned away and keeps probing)."""
    MAGIC = b"AEV2"

    def __init__(self, rank, world, addr=None, port=None, timeout=900.0):   # (ranks may reach their first collective minutes apart: each generates its streams first)
        self.rank, self.world = int(rank), int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = rendezvous_port() + PORT_SPAN if port is None else port
        token = job_token()
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
                        magic, tok, peer = struct.unpack(HS, _recv_exact(conn, HS_BYTES))
                        if magic != self.MAGIC or tok != token or not 0 < peer < self.world or peer in self.socks:
                            conn.close()        # (a stray connection, or a rank of another job whose port window overlaps ours)
                            continue
                        conn.sendall(struct.pack(HS, self.MAGIC, token, self.world))
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
                    conn.sendall(struct.pack(HS, self.MAGIC, token, self.rank))
                    magic, tok, w = struct.unpack(HS, _recv_exact(conn, HS_BYTES))
                    if magic == self.MAGIC and tok == token and w == self.world:     # (else: not my rank 0 -- keep probing)
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


REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXIT_RCCL_STUCK = 3


def _probe_main():
    """Child process of open_comm: the whole RCCL bring-up of this rank -- ncclGetUniqueId (rank 0), ncclCommInitRank, one all-reduce,
    destroy -- so that a bring-up that HANGS (a rank that cannot reach its peers blocks inside ncclCommInitRank for good) is a child
    the parent kills by PID, and the parent itself never has a thread stuck inside RCCL beside its GPU work.
    Protocol on the pipes: rank 0 prints "MADE <hex id>"; every child reads "UID <hex id>" from stdin; then "PROBE_OK" or
    "PROBE_ERR <why>"."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    try:
        if rank == 0:
            print("MADE " + Comm.unique_id().hex(), flush=True)
        uid = b""
        for ln in sys.stdin:
            if ln.startswith("UID "):
                uid = bytes.fromhex(ln.split()[1])
                break
        if len(uid) != ID_BYTES:
            raise RuntimeError("no ncclUniqueId from rank 0")
        ndev = N.device_count()
        c = Comm(rank, world, uid, int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1))
        got = float(c.allreduce([1.0], SUM)[0])
        c.close()
        if got != float(world):
            raise RuntimeError("all-reduce over %d ranks returned %g" % (world, got))
        print("PROBE_OK", flush=True)
    except Exception as e:      # noqa: BLE001 -- whatever it is, the parent must hear about it
        print("PROBE_ERR rank %d: %s" % (rank, str(e).replace("\n", " ")), flush=True)
        sys.exit(1)


class _Probe:
    """The parent's end of one probe child (see _probe_main)."""

    def __init__(self):
        code = "import sys; sys.path.insert(0, %r); from async_ev_cnn_amd import comm; comm._probe_main()" % REPO
        self.p = subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        self.buf = b""

    def line(self, prefixes, deadline):
        """the next stdout line that starts with one of `prefixes` (RCCL prints banners of its own), or None at the deadline / EOF"""
        fd = self.p.stdout.fileno()
        while True:
            while b"\n" in self.buf:
                ln, self.buf = self.buf.split(b"\n", 1)
                txt = ln.decode("utf-8", "replace").strip()
                if txt.startswith(tuple(prefixes)):
                    return txt
            left = deadline - time.time()
            if left <= 0 or not select.select([fd], [], [], left)[0]:
                return None
            part = os.read(fd, 65536)
            if not part:
                return None
            self.buf += part

    def send(self, text):
        try:
            self.p.stdin.write(text.encode())
            self.p.stdin.flush()
        except OSError:
            pass

    def end(self):
        """the child is done or given up on: killed by its PID if it still runs (never by pattern)"""
        if self.p.poll() is None:
            self.p.kill()
        try:
            self.p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pass
        for f in (self.p.stdin, self.p.stdout):
            try:
                f.close()
            except OSError:
                pass


def probe_rccl(tcp, timeout=240.0):
    """RCCL bring-up of all ranks in CHILD processes, agreed on over the star -> "" (every rank's child made a communicator and
    all-reduced over it) or the first rank's reason.  Call it BEFORE the rank touches the GPU (the children are ordinary child
    processes of a process that has not initialised HIP)."""
    deadline = time.time() + timeout
    pr, err = None, ""
    try:
        pr = _Probe()
    except OSError as e:
        err = "rank %d: probe process: %s" % (tcp.rank, e)
    uid_hex = b""
    if tcp.rank == 0 and pr is not None:
        ln = pr.line(("MADE ", "PROBE_ERR"), deadline)
        if ln and ln.startswith("MADE "):
            uid_hex = ln.split()[1].encode()
        else:
            err = ln[len("PROBE_ERR "):] if ln else "rank 0: ncclGetUniqueId did not return within %.0f s" % timeout
    uid_hex = tcp.bcast_bytes(uid_hex)
    if pr is not None and not err:
        if uid_hex:
            pr.send("UID %s\n" % uid_hex.decode())
            ln = pr.line(("PROBE_OK", "PROBE_ERR"), deadline)
            if ln is None:
                err = "rank %d: ncclCommInitRank did not return within %.0f s (probe process killed)" % (tcp.rank, timeout)
            elif ln.startswith("PROBE_ERR"):
                err = ln[len("PROBE_ERR "):]
        else:
            err = "rank %d: no ncclUniqueId from rank 0" % tcp.rank
    if pr is not None:
        pr.end()
    errs = [p.decode("utf-8", "replace") for p in tcp.allgather_bytes(err.encode()) if p]
    return errs[0] if errs else ""


def open_comm(device=None, rccl_timeout=240.0, log=None, probed=None):
    """The communicator of this job's ranks -> (comm, backend): an RCCL communicator (`Comm`, backend "rccl") if EVERY rank gets one,
    else the TCP star (`TcpComm`, backend "tcp: <why>") on every rank -- LOUDLY: comm.rccl_ranks is 0 then (the world size on
    success, from aev_comm_world), and bench.py prints it with "dist_backend_ok": false at the top level of its line.
    Order of events: the ranks connect the TCP star (plain sockets, job-token handshake); the RCCL bring-up is PROBED in child
    processes (probe_rccl: a bring-up that hangs is a child that gets killed, not a thread stuck inside RCCL in this process); only if
    every rank's probe succeeded does this process create its communicator -- rank 0's ncclUniqueId travels over the star -- under a
    watchdog: a rank whose ncclCommInitRank does not return within `rccl_timeout` although its probe did says so and EXITS non-zero
    (it cannot continue: nothing may share the GPU with a half-made communicator, and nothing is restarted or re-executed).
    `probed`: (tcp, reason) from a probe the caller has run already, before it touched the GPU (bench.py does)."""
    import threading
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = N.default_device() if device is None else device
    tcp, why = probed if probed is not None else (None, None)
    if tcp is None:
        tcp = TcpComm(rank, world)
        why = probe_rccl(tcp, rccl_timeout)

    def star(reason):
        if log:
            log("[aev comm] NO RCCL COMMUNICATOR (%s): control plane and final gather use the host-side TCP star; rccl_ranks = 0" % reason)
        tcp.rccl_ranks = 0
        return tcp, "tcp: " + reason

    if why:
        return star(why)
    uid = b""
    if rank == 0:
        try:
            uid = Comm.unique_id()
        except Exception as e:      # noqa: BLE001
            why = "rank 0: ncclGetUniqueId: %s" % e
    uid = tcp.bcast_bytes(uid)
    box = {}
    if len(uid) == ID_BYTES:
        def create():
            try:
                box["comm"] = Comm(rank, world, uid, device)
            except Exception as e:      # noqa: BLE001
                box["err"] = "rank %d: %s" % (rank, e)
        th = threading.Thread(target=create, daemon=True)
        th.start()
        th.join(rccl_timeout)
        if th.is_alive():
            msg = ("[aev comm] rank %d: ncclCommInitRank did not return within %.0f s although the probe process's did: this rank cannot "
                   "continue (a thread is stuck inside RCCL on its device) -- exiting with code %d" % (rank, rccl_timeout, EXIT_RCCL_STUCK))
            print(msg, file=sys.stderr, flush=True)
            os._exit(EXIT_RCCL_STUCK)
    else:
        box["err"] = why or "rank 0: no ncclUniqueId"
    errs = [p.decode("utf-8", "replace") for p in tcp.allgather_bytes(box.get("err", "").encode()) if p]
    if not errs and "comm" in box:
        tcp.close()
        c = box["comm"]
        c.rccl_ranks = int(N.lib().aev_comm_world(c.handle))
        return c, "rccl"
    if "comm" in box:       # (the init thread has returned on this rank: nothing of RCCL is left running)
        box["comm"].close()
    return star(errs[0])
