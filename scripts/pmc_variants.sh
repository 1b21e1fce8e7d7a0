#!/bin/bash
# Usage: pmc_variants.sh "NAME:FLAGS" ... -- per build variant, collect unit-busy counters for the
# integrate kernel (one rocprofv3 --pmc pass per group, short bench run).
cd /root/repo
GROUPS_=("SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
         "SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_CSN_BUSY SPI_RA_REQ_NO_ALLOC_CSN")
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  touch emfusion_amd/csrc/*.hip
  make -s -C emfusion_amd/csrc -j8 EXTRA="$flags" >/tmp/build_$name.log 2>&1 || { echo "$name build failed"; tail -5 /tmp/build_$name.log; continue; }
  echo "=== VARIANT $name"
  bash scripts/pmc_kernels.sh v_$name "${GROUPS_[@]}" 2>&1 | grep -E "integrate_batched|rc="
done
