"""Pretty-print a bench.py JSON line (dev helper)."""
import json, sys
d = json.load(open(sys.argv[1]))
print("value %.0f ev/s  ms/step %.4f  streams %d  speedup_vs_cpu %s" % (d['value'], d['ms_per_step'], d['config']['streams_per_gpu'], d.get('speedup_vs_cpu_baseline')))
for k, v in d['kernels'].items():
    print("%-16s %9.2f us  share %.3f  %s %s frac %s" % (k, v['us_per_launch'], v['share'], v.get('achieved'), v.get('unit'), v.get('frac')))
print(d['roofline'])
for l, v in (d.get('us_per_launch_by_layer') or {}).items():
    print(l, v)
