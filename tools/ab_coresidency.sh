#!/bin/bash
# Headline workload with the first-layer kernel at 1..4 workgroups per CU and the 64 x 64 halo kernel at 4 / 8 waves per workgroup
# (experiment build: tools/build_variant.py exp qnn_first_u8.hip,qnn_mfma_areg.hip -DQNN_EXPERIMENTS), 2 and 3 batches in flight:
# do first-layer(b+1) and B0(b) waves share the CUs when neither kernel's persistent grid takes every register?
export QNN_LIB=$PWD/quantizedneuralnetworks-keras-tensorflow_amd/csrc/variants/libqnn_exp.so
for nw in 8 4; do for bpc in 4 3 2 1; do for fl in 2 3; do
  QNN_HALO_NW=$nw QNN_U8_BPC=$bpc python bench.py --steps 30 --warmup 5 --inflight $fl --no-cpu-baseline --no-targets --no-alternatives 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('halo_nw=$nw u8_bpc=$bpc inflight=$fl', round(d['value']/1e6,2), 'M img/s', round(d['ms_per_step']*1e3,2), 'us', [round(k['ms']*1e3,1) for k in d.get('kernels',[])])"
done; done; done
