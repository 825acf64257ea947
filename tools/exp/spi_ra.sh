#!/bin/bash
# experiment: why workgroups are not resident -- SPI resource-allocation stall counters of the sweep kernels (separate --pmc passes)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
BOX=${1:-32}
for C in "SPI_RA_LDS_CU_FULL_CSN SPI_RA_BAR_CU_FULL_CSN" "SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN" "SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN" "SPI_RA_BULKY_CU_FULL_CSN SPI_RA_WVLIM_STALL_CSN" "SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN" "SPI_RA_TMP_STALL_CSN SPI_CSN_BUSY" "SPI_CSN_NUM_THREADGROUPS SPI_CSN_WAVE" "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES"; do
  D=/tmp/spi_$(echo $C | tr ' ' '_' | cut -c1-40)
  rm -rf $D
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -- python3 tools/prof_driver.py 512 $BOX 2 > $D.out 2>&1 || echo "pass ($C) failed: $(tail -1 $D.out)"
  for c in $C; do python3 tools/exp/pmc_sum.py $D $c march3; done
done
