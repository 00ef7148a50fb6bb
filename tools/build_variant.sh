#!/bin/bash
# Build librpt_hip.so with extra flags into rust-path-tracer_amd/lib/variants/NAME.so (A/B runs: RPT_HIP_LIB=... bench.py)
# usage: tools/build_variant.sh NAME -DRPT_X=1 ...      (SLP=1 in the environment: WITH the SLP vectorizer the Makefile switches off)
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/rust-path-tracer_amd/csrc
N="-fno-slp-vectorize"; [ "${SLP:-0}" = "1" ] && N=""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++20 -O3 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function $N -DRPT_BUILD_FINGERPRINT=\"$(python3 $ROOT/tools/source_fingerprint.py)+$NAME\" "$@" -shared -o $ROOT/rust-path-tracer_amd/lib/variants/$NAME.so $C/rpt_hip.hip $C/rpt_traverse.hip $C/rpt_comm.hip $C/rpt_lights.hip $C/rpt_bvh.hip -ldl
