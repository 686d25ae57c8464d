"""Experiment (latency mode): how long does the HOST need to submit a queued run of one stream, against the device's time?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import async_ev_cnn_amd as A
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K, W = 400, 24
layers = A.layers_dict(A.net.EFCN_LAYERS)
w = A.net.he_init_weights(layers, 1234)
net = A.EventNet(160, 224, layers, w, 'SAME', 0.1, 5e-5, S)
per = [bench.make_stream("ncaltech", 1234 + s, W + K, 200) for s in range(S)]
net.queue_upload([[per[s][t] for s in range(S)] for t in range(W + K)])
net.queue_run(0, W, True); net.sync()
t0 = time.perf_counter()
net.queue_run(W, K, True)
t1 = time.perf_counter()
net.sync()
t2 = time.perf_counter()
print("streams %d: queue_run returned after %.4f ms per step, the device was done after %.4f ms per step" % (S, 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K))
