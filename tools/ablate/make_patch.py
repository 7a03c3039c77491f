#!/usr/bin/env python3
"""Regenerates tools/ablate/scores_kernel_ablate.patch from the CURRENT kernel sources: the timing-only forks that earlier rounds
measured with (results wrong by construction), re-inserted into a scratch copy of mdir_amd/csrc and diffed against the tree.  The
shipped sources carry none of them.

    -DMDX_ABL_NOLOAD       the LDS ring is filled once and never again: no LDS-DMA beside the MFMAs        (tools/scores_where.sh)
    -DMDX_ABL_NOLDSREAD    operands are read for the first chunk and reused: no ds_read beside the MFMAs
    -DMDX_ABL_SAME_ROWS    every workgroup streams the same few rows: the shard comes from the L2 instead of HBM
    -DMDX_ABL_NO_EPILOGUE  what the epilogue costs (every accumulator stays alive in one sum)
    -DMDX_SHARD_BLOCKED    blocks of 16 row tiles with the k-block as the slow index
    -DMDX_XCD_BLOCKS=0     workgroup id = row block (no XCD-contiguous ranges)
    -DMDX_GRAM_ABL=5 / -DMDX_GRAM_PLAIN_MAP   the f64 GEMM's loaders issue nothing inside the loop / workgroup id = tile position
(The in-kernel time stamps of round 2 and the split kernels' ABL template argument are in the history: git show 7e802e4:mdir_amd/csrc/.)

    python tools/ablate/make_patch.py        # rewrites the patch; tools/scores_where.sh applies it to a scratch copy
"""
import difflib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "mdir_amd", "csrc")


def edit(text, pairs):
    for a, b in pairs:
        assert text.count(a) == 1, a[:80]
        text = text.replace(a, b)
    return text


KERNEL = [
    ("__host__ __device__ __forceinline__ int64_t shard_tile(int64_t rt, int64_t kb, int64_t KB) { return rt * KB + kb; }\n",
     "#ifdef MDX_SHARD_BLOCKED\nconstexpr int SHARD_BLOCK = 16;\n__host__ __device__ __forceinline__ int64_t shard_tile(int64_t rt, int64_t kb, int64_t KB)\n{\n"
     "    return ((rt / SHARD_BLOCK) * KB + kb) * SHARD_BLOCK + (rt % SHARD_BLOCK);\n}\n#else\n"
     "__host__ __device__ __forceinline__ int64_t shard_tile(int64_t rt, int64_t kb, int64_t KB) { return rt * KB + kb; }\n#endif\n"),
    ("    const unsigned per = nblocks / 8, rem = nblocks % 8;",
     "#if defined(MDX_XCD_BLOCKS) && MDX_XCD_BLOCKS == 0\n    return id;\n#endif\n    const unsigned per = nblocks / 8, rem = nblocks % 8;"),
    ("                if constexpr (RM) {\n                    const int64_t row = (rt_wg + tile) * TILE_ROWS + (lane & 15);",
     "#ifdef MDX_ABL_SAME_ROWS\n                if (true) {\n                    src[t] = db + shard_tile((int64_t)(blockIdx.x % 16) * CW * R + tile, kbc, KB) * 64 + lane;\n                } else\n#endif\n"
     "                if constexpr (RM) {\n                    const int64_t row = (rt_wg + tile) * TILE_ROWS + (lane & 15);"),
    ("            if (c + NSTAGE - 1 < nchunks) issue(c + NSTAGE - 1);            // refill the slot of stage c-1\n",
     "#ifndef MDX_ABL_NOLOAD\n            if (c + NSTAGE - 1 < nchunks) issue(c + NSTAGE - 1);            // refill the slot of stage c-1\n#endif\n"),
    ("        auto read_first = [&](const f32x4 *slot, int kb) __attribute__((always_inline)) {\n",
     "#ifdef MDX_ABL_NOLDSREAD\n        bool abl_read = true;\n#else\n        constexpr bool abl_read = true;\n#endif\n"
     "        auto read_first = [&](const f32x4 *slot, int kb) __attribute__((always_inline)) {\n            if (!abl_read) return;\n"),
    ("        auto read_left = [&](const f32x4 *slot, int kb) __attribute__((always_inline)) {\n",
     "        auto read_left = [&](const f32x4 *slot, int kb) __attribute__((always_inline)) {\n            if (!abl_read) return;\n"),
    ("if (t == 3 && q != 0 && nslot) {", "if (t == 3 && q != 0 && nslot && abl_read) {"),
    ("            read_left(slot, KC - 1);\n", "            read_left(slot, KC - 1);\n#ifdef MDX_ABL_NOLDSREAD\n            if (c >= 1) abl_read = false;\n#endif\n"),
    ("    static_assert((QT * 16 + QR * 8) * LDW * 4 <= NSTAGE * STAGE_TILES * 1024, \"output staging must fit in the ring\");\n",
     "#ifdef MDX_ABL_NO_EPILOGUE\n    {\n        f32x4 tot = accl;\n#pragma unroll\n        for (int r = 0; r < R; ++r)\n#pragma unroll\n            for (int q = 0; q < QT; ++q) tot += acc[r][q];\n"
     "        out[(int64_t)(lane % 64) * n + rt_wg * TILE_ROWS + wave] = tot[0] + tot[1] + tot[2] + tot[3];\n        return;\n    }\n#endif\n"
     "    static_assert((QT * 16 + QR * 8) * LDW * 4 <= NSTAGE * STAGE_TILES * 1024, \"output staging must fit in the ring\");\n"),
]
GRAM = [
    ("    int64_t L = (int64_t)x * per + (x < rem ? x : rem) + k;            // position in the tile sequence\n",
     "    int64_t L = (int64_t)x * per + (x < rem ? x : rem) + k;            // position in the tile sequence\n#ifdef MDX_GRAM_PLAIN_MAP\n    L = id;\n#endif\n"),
    ("            if (c + NSTAGE - 1 < nsteps) issue(c + NSTAGE - 1);\n",
     "#if !defined(MDX_GRAM_ABL) || MDX_GRAM_ABL != 5\n            if (c + NSTAGE - 1 < nsteps) issue(c + NSTAGE - 1);\n#endif\n"),
]

out = []
for name, pairs in (("mdx_scores_kernel.h", KERNEL), ("mdx_gram.hip", GRAM)):
    old = open(os.path.join(CSRC, name)).read()
    new = edit(old, pairs)
    out += list(difflib.unified_diff(old.splitlines(True), new.splitlines(True), "a/" + name, "b/" + name))
with open(os.path.join(ROOT, "tools", "ablate", "scores_kernel_ablate.patch"), "w") as f:
    f.writelines(out)
print("wrote tools/ablate/scores_kernel_ablate.patch (%d lines)" % len(out))
