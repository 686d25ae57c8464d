#!/bin/bash
O=gpurun_out/r2_dbg; mkdir -p $O; rm -f $O/segv*.txt
AEV_LA_STREAM3=1 AEV_SEGV_TRACE=$PWD/$O/segv_s3.txt python -m pytest tests -m gpu -x -q -p no:faulthandler > $O/full_s3.log 2>&1; echo "stream3 rc=$?"; tail -1 $O/full_s3.log | cut -c1-100
AEV_NO_OVERLAP=1 AEV_LOOKAHEAD=1 AEV_SEGV_TRACE=$PWD/$O/segv_noov.txt python -m pytest tests -m gpu -x -q -p no:faulthandler > $O/full_noov.log 2>&1; echo "no-overlap + LA rc=$?"; tail -1 $O/full_noov.log | cut -c1-100
AEV_SEGV_TRACE=$PWD/$O/segv_def.txt python -m pytest tests -m gpu -x -q -p no:faulthandler > $O/full_def.log 2>&1; echo "default rc=$?"; tail -1 $O/full_def.log | cut -c1-100
