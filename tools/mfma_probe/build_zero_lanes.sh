#!/bin/bash
# The builds of gram128_zero_lanes.hip that profiles/probes_r05.md section 4 tabulates (tools/profile_round.sh runs them):
#   tools/mfma_probe/build_zero_lanes.sh ; then on the GPU box ./gram128_zero_lanes_<name> 512 16384 2000 600
cd "$(dirname "$0")"
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3"
P="-mllvm --amdgpu-mfma-padding-ratio=100"
b() { n=$1; shift; $H "$@" -o gram128_zero_lanes_$n gram128_zero_lanes.hip 2>&1 | grep -E " error" ; }
b base
b n0 -DNO_SUBDIAG_SKIP                                   # the round-4 kernel with all four blocks computed (higher rate)
b one -DWG_PER_CU=1                                      # one workgroup per CU
b p4 -DNO_SUBDIAG_SKIP -DPARTNER=4                       # control of the stand-in grid: the kernel in both slots of a CU
b p1 -DNO_SUBDIAG_SKIP -DPARTNER=1                       # second workgroup of a CU = MFMAs only
b p2 -DNO_SUBDIAG_SKIP -DPARTNER=2                       # ... LDS traffic only
b p3 -DNO_SUBDIAG_SKIP -DPARTNER=3                       # ... global loads only
b il -DINTERLEAVE                                        # MFMAs product-major
b g2 -DINTERLEAVE -DGAP=2                                # ... with s_sleep 2 between the groups (amplifier)
b g2one -DINTERLEAVE -DGAP=2 -DWG_PER_CU=1
b pad -DNO_SUBDIAG_SKIP $P                               # LLVM's MFMA padding (s_nop between MFMAs: the other amplifier)
b n0s -DNO_SUBDIAG_SKIP -DSCALAR_FOLD                    # the fold as v_fma_f32 instead of the compiler's v_pk_fma_f32
b g2s -DINTERLEAVE -DGAP=2 -DSCALAR_FOLD
b pads -DNO_SUBDIAG_SKIP -DSCALAR_FOLD $P
b g2p -DINTERLEAVE -DGAP=2 -DPK_ASM_FOLD                 # the fold as inline-asm v_pk_fma_f32 (same pinning as SCALAR_FOLD)
b padp -DNO_SUBDIAG_SKIP -DPK_ASM_FOLD $P
b g2pc -DINTERLEAVE -DGAP=2 -DPK_ASM_FOLD -DPK_COPY_FIRST # ... reading copies of the accumulators
b padpc -DNO_SUBDIAG_SKIP -DPK_ASM_FOLD -DPK_COPY_FIRST $P
b g2ma -DINTERLEAVE -DGAP=2 -DPK_ASM_FOLD -DPK_MUL_ADD   # v_pk_mul_f32 + v_pk_add_f32
b padma -DNO_SUBDIAG_SKIP -DPK_ASM_FOLD -DPK_MUL_ADD $P
ls gram128_zero_lanes_* | wc -l
