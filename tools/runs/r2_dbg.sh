#!/bin/bash
O=gpurun_out/r2_dbg; mkdir -p $O
timeout -k 10 300 python -X faulthandler -m pytest tests -m gpu -x -v > $O/full_default.log 2>&1; echo "default rc=$?"; grep -n "PASSED\|FAILED\|Segmentation" $O/full_default.log | tail -4
AEV_LOOKAHEAD=0 timeout -k 10 300 python -m pytest tests -m gpu -x -q > $O/full_la0.log 2>&1; echo "LOOKAHEAD=0 rc=$?"; tail -2 $O/full_la0.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -v -k "multi_stream_queue_graph or many_streams_second" > $O/two.log 2>&1; echo "two tests rc=$?"; tail -3 $O/two.log
