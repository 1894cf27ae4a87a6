#!/bin/bash
# on the GPU box: tools/run_stamps.sh VARIANT -> phase stamps of the attention backward with that variant library
L=visiontransformer_amd/libdswin_hip.so
cp $L /tmp/_o.so; cp visiontransformer_amd/variants/libdswin_$1.so $L
timeout -k 10 120 python tools/bwd_stamps.py; rc=$?
cp /tmp/_o.so $L; exit $rc
