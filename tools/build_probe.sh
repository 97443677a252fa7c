#!/bin/bash
# Diagnostic builds of the library for tools/clock_probe.py: tools/probe/libcoda_hip_probe[_<tag>].so
# usage: tools/build_probe.sh [tag "-DFLAG ..."]...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/coda_amd/csrc"
mkdir -p "$ROOT/tools/probe" /tmp/probe_obj
build() {
  tag=$1; shift
  for f in abi_misc elementwise gemm_bf16 flash_attn lmhead_confidence confidence_logits sampler; do
    if [ "$f" = gemm_bf16 ] || [ ! -f /tmp/probe_obj/$f.o ]; then
      hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -I. -Wno-unused-function -ffp-contract=off -DCODA_CLOCK_PROBE $@ -c $f.hip -o /tmp/probe_obj/$f$([ "$f" = gemm_bf16 ] && echo "_$tag").o
    fi
  done
  hipcc -shared -fPIC --offload-arch=gfx950 /tmp/probe_obj/abi_misc.o /tmp/probe_obj/elementwise.o /tmp/probe_obj/gemm_bf16_$tag.o /tmp/probe_obj/flash_attn.o \
     /tmp/probe_obj/lmhead_confidence.o /tmp/probe_obj/confidence_logits.o /tmp/probe_obj/sampler.o -o "$ROOT/tools/probe/libcoda_hip_probe_$tag.so"
}
if [ $# -gt 0 ]; then
  while [ $# -gt 0 ]; do build $1 $2; shift 2; done
  exit 0
fi
build base
build nolgkm -DCODA_DIAG_NO_LGKM
build nosync -DCODA_DIAG_NO_SYNC
build nobarrier -DCODA_DIAG_NO_BARRIER
build nodma -DCODA_DIAG_NO_DMA
build bare "-DCODA_DIAG_NO_LGKM -DCODA_DIAG_NO_SYNC -DCODA_DIAG_NO_DMA"
build hotk -DCODA_DIAG_HOTK
build novmcnt -DCODA_DIAG_NO_VMCNT
