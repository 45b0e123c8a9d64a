"""Developer aid: patch csrc/conv_bww_mfma.hip IN PLACE with per-phase cycle counters of one block of the producer/consumer
variant, printed by the launcher when MFVI_PROF is set.  Columns per wave: T0 prologue, T1 prefetch issue, T2 wait B1,
T3 stage (registers -> LDS), T4 wait B2, T5 MFMA, T6 (unused), T7 reduction + slab store.  Restore with `git checkout`."""
import sys
p = sys.argv[1] if len(sys.argv) > 1 else 'mfvi-dip-mia_amd/csrc/conv_bww_mfma.hip'
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b, 1)
rep("int tiles_per_block, int ci_groups, int nx, int ny, int nz)\n{", "int tiles_per_block, int ci_groups, int nx, int ny, int nz, long long* prof)\n{\n    long long T[8] = {0,0,0,0,0,0,0,0}; long long tc = clock64();\n#define TICK(i) { const long long n_ = clock64(); T[i] += n_ - tc; tc = n_; }\n")
rep('''            if (tile_begin < tile_end) pfetch(tile_begin);
            for (int tile = tile_begin; tile < tile_end; ++tile) {
                if (tile > tile_begin) __syncthreads();                  // (B1) consumers are done with the previous tile
                pstage();
                __syncthreads();                                         // (B2) tile published
                if (tile + 1 < tile_end) pfetch(tile + 1);
            }''', '''            TICK(0)
            if (tile_begin < tile_end) pfetch(tile_begin);
            TICK(1)
            for (int tile = tile_begin; tile < tile_end; ++tile) {
                if (tile > tile_begin) __syncthreads();                  // (B1) consumers are done with the previous tile
                TICK(2)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TICK(6)
                pstage();
                TICK(3)
                __syncthreads();                                         // (B2) tile published
                TICK(4)
                if (tile + 1 < tile_end) pfetch(tile + 1);
                TICK(1)
            }''')
rep('''            __syncthreads();                                             // (S0)
            for (int tile = tile_begin; tile < tile_end; ++tile) {
                if (tile > tile_begin) __syncthreads();                  // (B1)
                __syncthreads();                                         // (B2)
                mfma_tile(tile);
            }''', '''            __syncthreads();                                             // (S0)
            TICK(0)
            for (int tile = tile_begin; tile < tile_end; ++tile) {
                if (tile > tile_begin) __syncthreads();                  // (B1)
                TICK(2)
                __syncthreads();                                         // (B2)
                TICK(4)
                mfma_tile(tile);
                TICK(5)
            }''')
rep("    // ---- sum the MFMA waves through LDS", "    TICK(6)\n    // ---- sum the MFMA waves through LDS")
rep("""    if (do_bias && t < cot) o[(long long)Cout * Cin * KK + co0 + t] = (s_db[t] + s_db[16 + t]) + (s_db[32 + t] + s_db[48 + t]);
}""", """    if (do_bias && t < cot) o[(long long)Cout * Cin * KK + co0 + t] = (s_db[t] + s_db[16 + t]) + (s_db[32 + t] + s_db[48 + t]);
    TICK(7)
    if (prof && bx == 1 && by == 0 && k == 3 && lane == 0) for (int i = 0; i < 8; ++i) prof[wv * 8 + i] = T[i];
}""")
rep("ci_groups, strips, co_tiles * ci_groups, n_samples);            \\", "ci_groups, strips, co_tiles * ci_groups, n_samples, prof);      \\\n        if (prof) { long long h[64]; (void)hipStreamSynchronize(st); (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost); fprintf(stderr, \"BWW ks %d s %d nb %d nt %d spec %d strips %d tpb %d\\n\", KS_, S_, NB_, NT_, (int)SP_, strips, tpb); for (int w = 0; w < NT_ / 64; ++w) { fprintf(stderr, \"wave %d:\", w); for (int i = 0; i < 8; ++i) fprintf(stderr, \" %lld\", h[w * 8 + i]); fprintf(stderr, \"\\n\"); } } \\")
rep("    int cfg = g.tune[2] ? g.tune[2] : env_tune_w();", "    static long long* prof = [] { long long* p = nullptr; if (getenv(\"MFVI_PROF\")) { (void)hipMalloc((void**)&p, 64 * 8); (void)hipMemset(p, 0, 64 * 8); } return p; }();\n    int cfg = g.tune[2] ? g.tune[2] : env_tune_w();")
rep("#include <cstdlib>", "#include <cstdlib>\n#include <cstdio>")
open(p, 'w').write(s)
