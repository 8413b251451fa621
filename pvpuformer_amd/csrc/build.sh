#!/bin/bash
# Builds libvpu_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-value"
# `build.sh diag`: the diagnostic library libvpu_hip_diag.so (-DVPU_DIAG: time-stamp code in the K2 / K4P GEMM kernels, read by
# tools/k2_stamps.py and tools/k4_drift.py through VPU_LIB_DIAG=1).  The product library has neither the stamp pointer nor the code.
OUT=libvpu_hip.so; BUILD=build
if [ "$1" = "diag" ]; then FLAGS="$FLAGS -DVPU_DIAG -DVPU_LAB"; OUT=libvpu_hip_diag.so; BUILD=build_diag; fi
# `build.sh x`: an EXPERIMENT library libvpu_hip_x.so (same sources, VPU_X_FLAGS added to every file, VPU_X_<file> to one) for a
# same-box A/B against the product library: a process loads it with VPU_LIB_FILE=libvpu_hip_x.so (pvpuformer_amd/_lib.py).
if [ "$1" = "x" ]; then FLAGS="$FLAGS $VPU_X_FLAGS"; OUT=libvpu_hip_x.so; BUILD=build_x; fi
mkdir -p $BUILD
pids=()
for f in gemm gemm_k5 attention rowops spatial prompt loss optim; do
  # attention.hip: MFMA results straight into VGPRs (the softmax consumes every score tile with VALU instructions; with the
  # accumulator-register form hipcc copies each tile through v_accvgpr_read and the kernels drop to one wave per SIMD)
  EXTRA=""; [ $f = attention ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"
  # gemm_k5.hip: no SLP vectorisation -- its epilogue arithmetic runs BESIDE the partner wave's MFMAs, where a packed
  # v_pk_fma_f32 / v_pk_mul_f32 costs the issue port ~4x what two plain v_fma_f32 do (MI355X_MICROARCH.md, cycle constants)
  [ $f = gemm_k5 ] && EXTRA="-fno-slp-vectorize"
  # loss.hip: the same flag -- the pixel pass of p2cl_up is instruction-bound and the packed form costs two v_mov per v_pk_mul_f32
  # (252 -> 245 us, same bits)
  [ $f = loss ] && EXTRA="-fno-slp-vectorize"
  if [ "$1" = "x" ]; then v="VPU_X_$f"; EXTRA="$EXTRA ${!v}"; fi
  $HIPCC $FLAGS $EXTRA -c $f.hip -o $BUILD/$f.o &
  pids+=($!)
done
$HIPCC $FLAGS -c common.cpp -o $BUILD/common.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
# link beside the target and rename: a rename is atomic, so a process that polls for the library (bench.py, ranks > 0) or
# dlopens it while another one builds sees the old file or the complete new one, never a half-written ELF
TMP=../$OUT.tmp.$$
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $TMP $BUILD/*.o
mv -f $TMP ../$OUT
echo "built $(cd .. && pwd)/$OUT"
