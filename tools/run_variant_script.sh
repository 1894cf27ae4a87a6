#!/bin/bash
# on the GPU box: tools/run_variant_script.sh VARIANT SCRIPT [args] -> runs the script with that variant library in place
L=visiontransformer_amd/libdswin_hip.so
cp $L /tmp/_o.so; cp visiontransformer_amd/variants/libdswin_$1.so $L; shift
timeout -k 10 150 python "$@"; rc=$?
cp /tmp/_o.so $L; exit $rc
