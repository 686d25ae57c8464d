"""Experiment (latency mode): do the step graphs of G independent ONE-stream handles run side by side on the device?
If they do, G handles finish K steps each in about the time one handle needs -- graph replays on different HIP streams overlap,
and what serialises the two halves of a few-streams step is the executor's handling of branches INSIDE one graph."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import async_ev_cnn_amd as A
import bench

K, W = 200, 20
layers = A.layers_dict(A.net.EFCN_LAYERS)
w = A.net.he_init_weights(layers, 1234)
for G in (1, 2, 3, 4):
    nets = []
    for g in range(G):
        net = A.EventNet(160, 224, layers, w, 'SAME', 0.1, 5e-5, 1)
        per = bench.make_stream("ncaltech", 1234 + g, W + K, 200)
        net.queue_upload([[per[t]] for t in range(W + K)])
        nets.append(net)
    for net in nets:
        net.queue_run(0, W, True)
    for net in nets:
        net.sync()
    t0 = time.perf_counter()
    for net in nets:
        net.queue_run(W, K, True)
    for net in nets:
        net.sync()
    dt = time.perf_counter() - t0
    print("handles=%d: %.4f ms per step of each handle (%.4f ms per step and handle)" % (G, 1e3 * dt / K, 1e3 * dt / K / G))
    del nets

# the same with one host thread per handle (ctypes releases the GIL): is the replay bound by the host's submission of the graphs?
import threading
for G in (2, 4):
    nets = []
    for g in range(G):
        net = A.EventNet(160, 224, layers, w, 'SAME', 0.1, 5e-5, 1)
        per = bench.make_stream("ncaltech", 1234 + g, W + K, 200)
        net.queue_upload([[per[t]] for t in range(W + K)])
        net.queue_run(0, W, True)
        net.sync()
        nets.append(net)
    def work(net):
        net.queue_run(W, K, True)
        net.sync()
    th = [threading.Thread(target=work, args=(net,)) for net in nets]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("handles=%d, a host thread each: %.4f ms per step of each handle" % (G, 1e3 * dt / K))
    del nets
