#!/bin/bash
# Build librpt_hip.so with extra flags into rust-path-tracer_amd/lib/variants/NAME.so (A/B runs: RPT_HIP_LIB=... bench.py)
# usage: tools/build_variant.sh NAME -DRPT_X=1 ...      (SLP=1 in the environment: every translation unit WITH the SLP vectorizer)
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/rust-path-tracer_amd/csrc
O=$ROOT/rust-path-tracer_amd/lib/variants/obj_$NAME
mkdir -p $O
F="--offload-arch=gfx950 -std=c++20 -O3 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function"
N="-fno-slp-vectorize"; [ "${SLP:-0}" = "1" ] && N=""
/opt/rocm/bin/hipcc $F $N -DRPT_BUILD_FINGERPRINT=\"$(python3 $ROOT/tools/source_fingerprint.py)+$NAME\" "$@" -c -o $O/rpt_hip.o $C/rpt_hip.hip && \
/opt/rocm/bin/hipcc $F $N "$@" -c -o $O/rpt_comm.o $C/rpt_comm.hip && \
/opt/rocm/bin/hipcc $F "$@" -c -o $O/rpt_kernels_slp.o $C/rpt_kernels_slp.hip && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/rust-path-tracer_amd/lib/variants/$NAME.so $O/rpt_hip.o $O/rpt_comm.o $O/rpt_kernels_slp.o -ldl
rm -rf $O
