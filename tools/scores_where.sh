#!/bin/bash
# Where does the exact similarity kernel's idle fifth come from?  Timing-only builds of the library (results wrong): no LDS-DMA
# after the ring's first fill (-DMDX_ABL_NOLOAD), no ds_read after the first chunk (-DMDX_ABL_NOLDSREAD), both; each timed on
# gaussian unit rows and on all-zero operands by tools/scores_pipe_probe.py (the 8-consumer rows of the profile: library and probe
# of commit 16ee919, MDX_SCORES_CW8 / PROBE_CW8).  The variant libraries are built here when they are not there (~40 s each).
# Round 6: the shipped kernel sources carry no timing forks any more; they are tools/ablate/scores_kernel_ablate.patch, applied
# here to a SCRATCH copy of mdir_amd/csrc (never to the tree).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/where_r05; rm -rf $OUT; mkdir -p $OUT
ABL=/tmp/mdx_ablate_src; rm -rf $ABL; mkdir -p $ABL/mdir_amd; cp -r $R/mdir_amd/csrc $ABL/mdir_amd/csrc; cp -r $R/include $ABL/include
(cd $ABL/mdir_amd/csrc && patch -p1 < $R/tools/ablate/scores_kernel_ablate.patch) || exit 1
SRC="mdx_index.hip mdx_rank.hip mdx_pool.hip mdx_trunk.hip mdx_jpeg.hip mdx_comm.hip mdx_gram.hip mdx_conv.hip mdx_clahe.hip"
for v in NOLOAD NOLDSREAD "NOLOAD -DMDX_ABL_NOLDSREAD"; do
  n=$(echo $v | tr -d ' -' | sed 's/DMDX_ABL_//')
  [ -f $R/mdir_amd/libmdx_abl_$n.so ] || (cd $ABL/mdir_amd/csrc && /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result -DMDX_ABL_$v -shared -o $R/mdir_amd/libmdx_abl_$n.so $SRC)
done
for v in "" _abl_NOLOAD _abl_NOLDSREAD _abl_NOLOADNOLDSREAD; do
  MDIR_AMD_LIB=$R/mdir_amd/libmdx$v.so timeout 300 python3 $R/tools/scores_pipe_probe.py 2 > $OUT/probe$v.log 2>&1
  echo "== libmdx$v.so"; grep round $OUT/probe$v.log | sed 's/bit-equal.*//'
done
