"""One stream, B events per step (uniform pixels, as bench.py --sweep-batch): ms per step of a queued graph run (dev helper).
    python tools/exp_small_batches.py [B ...]        (under rocprofv3 --kernel-trace for the timeline of tools/timeline.py)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    eng = bench.Engine()
    for B in [int(v) for v in sys.argv[1:]] or [1, 20, 200]:
        W, K = 8, 64
        net = eng.make_net(1, 0)
        steps = bench.make_stream("uniform", 1234, W + K, B)
        net.queue_upload([[b] for b in steps])
        net.queue_run(0, W, True)
        net.sync()
        t0 = time.perf_counter()
        net.queue_run(W, K, True)
        net.sync()
        dt = time.perf_counter() - t0
        print("B = %5d: %.1f us per step, %.1f events/s" % (B, 1e6 * dt / K, B * K / dt), flush=True)
        del net


if __name__ == "__main__":
    main()
