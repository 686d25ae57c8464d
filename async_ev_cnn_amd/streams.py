"""Event-stream plumbing around the hot path: synthetic stream generators, the reference's
sample -> event-batch rule (src/libs/runner.py:23-33,65-72), stream sharding over ranks and the gather of
predictions (one RCCL all-gather over xGMI; the path itself never communicates)."""
import math

import numpy as np


# ------------------------------------------------------------------------------------------------ feeder
def center_crop_events(x, y, ts, new_shape):
    """Event half of utils.center_crop (src/libs/utils.py:11-26), quirk included: the reference derives the
    x-window offset from the y extent and vice versa (`new_top` from x, `new_left` from y, utils.py:17-21).
    Returns (x, y, ts) of the events inside, shifted so that their minimum is 0."""
    new_h, new_w = new_shape
    new_top = (int(x.max()) - int(x.min()) - new_w) // 2
    new_left = (int(y.max()) - int(y.min()) - new_h) // 2
    inside = (x >= new_left) & (x < new_left + new_w) & (y >= new_top) & (y < new_top + new_h)
    nx, ny, nts = x[inside].copy(), y[inside].copy(), ts[inside].copy()
    nx -= nx.min()
    ny -= ny.min()
    return nx, ny, nts


def sample_to_batches(x, y, ts, frame_hw, example_hw=None, batch_event_size=200, batch_event_usec=None):
    """Runner.data_transform + the batching rule of Runner.run (runner.py:23-33,65-72): ts -= ts[0]; center-crop
    if the frame differs from the example; stack (y,x,ts); split into ~batch_event_size chunks with
    np.array_split (or by time bins of batch_event_usec).  Returns a list of int32 [n,3] arrays."""
    ts = ts - ts[0]
    if example_hw is not None and tuple(example_hw) != tuple(frame_hw):
        x, y, ts = center_crop_events(x, y, ts, frame_hw)
    ev = np.stack([y, x, ts], axis=-1).astype(np.int32)
    if batch_event_usec is not None:
        bins = np.arange(0, ev[-1, 2], batch_event_usec)
        ids = np.digitize(ev[:, 2], bins)
        split = np.where(ids[:-1] != ids[1:])[0] + 1
        return [b for b in np.array_split(ev, split, axis=0) if len(b)]
    nb = int(math.ceil(ev.shape[0] / float(batch_event_size)))
    return [b for b in np.array_split(ev, nb, axis=0) if len(b)]


# ------------------------------------------------------------------------------------------------ synthetic streams
def ncaltech_like_sample(seed, sensor_hw=(180, 240), duration_us=300000, n_events=100000, n_edges=24, leg_us=None):
    """S-ncaltech of SURVEY.md 8d: K straight edge segments translating along a 3-leg triangular saccade
    (N-Caltech101 was recorded by moving the sensor in three 100 ms legs); an event = a point on a segment at its
    time + N(0, 1 px) jitter; timestamps uniform over the duration.  `leg_us`: duration of one leg (default: a third of the
    sample, i.e. 100 ms for the 300 ms standard sample); a longer stream walks the closed triangle again and again, so its
    statistics do not depend on its length.  Returns (x, y, ts) sorted by ts."""
    rs = np.random.RandomState(seed)
    H, W = sensor_hw
    cx, cy = rs.uniform(30, W - 30, n_edges), rs.uniform(30, H - 30, n_edges)
    ang, half = rs.uniform(0, np.pi, n_edges), rs.uniform(8, 40, n_edges)
    legs = np.array([[1.0, 0.6], [-1.0, 0.6], [0.0, -1.2]]) * 12.0  # px displacement per leg (closed triangle)
    leg_len = duration_us / 3.0 if leg_us is None else float(leg_us)
    ts = np.sort(rs.randint(0, duration_us, n_events)).astype(np.int64)
    cyc = (ts // leg_len).astype(np.int64)
    leg = cyc % 3 if leg_us is not None else np.minimum(cyc, 2)
    frac = (ts - cyc * leg_len) / leg_len if leg_us is not None else (ts - leg * leg_len) / leg_len
    start = np.concatenate([[np.zeros(2)], np.cumsum(legs, 0)[:2]])  # position at the start of each leg
    pos = start[leg] + legs[leg] * frac[:, None]
    e = rs.randint(0, n_edges, n_events)
    t = rs.uniform(-1, 1, n_events) * half[e]
    x = cx[e] + np.cos(ang[e]) * t + pos[:, 0] + rs.randn(n_events)
    y = cy[e] + np.sin(ang[e]) * t + pos[:, 1] + rs.randn(n_events)
    xi, yi = np.rint(x).astype(np.int64), np.rint(y).astype(np.int64)
    ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
    return xi[ok], yi[ok], ts[ok]


def ncaltech_like_batches(seed, frame_hw=(160, 224), batch_event_size=200, **kw):
    """A synthetic N-Caltech101-shaped stream, cropped and batched exactly as the reference's runner would."""
    x, y, ts = ncaltech_like_sample(seed, **kw)
    return sample_to_batches(x, y, ts, frame_hw, example_hw=kw.get("sensor_hw", (180, 240)),
                             batch_event_size=batch_event_size)


def uniform_batches(seed, frame_hw, n_steps, batch_event_size, period_us, t0=0):
    """S-uniform of SURVEY.md 8d: per step B events, uniform pixels, ts = sort(randint(0, T, B)) + t_prev.
    period_us == B gives 1 event / us = 1 Mevent/s."""
    rs = np.random.RandomState(seed)
    H, W = frame_hw
    out, prev = [], t0
    for _ in range(n_steps):
        ts = np.sort(rs.randint(0, max(period_us, 1), batch_event_size)) + prev
        out.append(np.stack([rs.randint(0, H, batch_event_size), rs.randint(0, W, batch_event_size), ts], -1)
                   .astype(np.int32))
        prev = int(ts.max()) + 1
    return out


# ------------------------------------------------------------------------------------------------ sharding
def shard_streams(n_total, rank, world_size):
    """Round-robin assignment of independent streams to ranks (stream i -> rank i mod world): the path is
    embarrassingly parallel over streams, no data-path collective."""
    return list(range(rank, n_total, world_size))


def gather_predictions(local, group=None, force=False):
    """All-gather of each rank's prediction tensor [S_local, h, w, c] -> [world, S_max, h, w, c] plus the per-rank
    stream counts.  `local` is a torch tensor on the rank's device (RCCL) or on the CPU (gloo).  One direct all-gather:
    every peer is one xGMI hop away and the message is 15 KB per stream, i.e. latency-bound.

    `shard_streams` hands ranks unequal stream counts whenever n_total % world != 0; a collective needs equal shapes on
    every rank, so the counts are exchanged first (one tiny all-gather) and every rank pads its tensor to the largest
    count.  Returns (stacked, counts): rows [r, counts[r]:] of `stacked` are padding (zeros)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return local.unsqueeze(0), [int(local.shape[0])]     # (force: run the collectives even with one rank -- a rehearsal)
    world = dist.get_world_size(group)
    mine = torch.tensor([int(local.shape[0])], dtype=torch.int64, device=local.device)
    cnts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(cnts, mine, group=group)
    counts = [int(c.item()) for c in cnts]
    smax = max(counts)
    send = local.contiguous()
    if send.shape[0] < smax:
        pad = torch.zeros((smax - send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], 0)
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send, group=group)
    return torch.stack(parts, 0), counts
