#!/bin/bash
# A/B SQ counters of the line-extraction kernels for the product library and variants (tools/build_variant.sh): tools/pmc_lsd_ab.sh NAME...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in product "$@"; do
  lib=""; [ "$v" != product ] && lib=$R/structure-slam-pointline_amd/lib/variants/$v.so
  for pass in 1 2; do
    if [ $pass = 1 ]; then C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; else C="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; fi
    rm -rf $O/$v.$pass
    (cd $R && SSLAM_LIB=$lib timeout 300 rocprofv3 --pmc $C -d $O/$v.$pass -- python tools/lsd_only.py 6144 8 1 > $O/$v.$pass.log 2>&1)
    (cd $R && python tools/rocpd_pmc_summary.py $O/$v.$pass $O/$v.$pass.txt | grep "k_lsd_regions" | awk -v v=$v '{printf "%s %s %s %.4g\n", v, $1, $2, $4}')
    rm -rf $O/$v.$pass
  done
done
