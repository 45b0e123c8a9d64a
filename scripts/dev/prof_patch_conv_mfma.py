"""Developer aid: patch csrc/conv_mfma.hip IN PLACE with per-phase cycle counters (clock64 around prefetch / store / weight
sampling / barrier waits / MFMA / epilogue of one block) printed by the launcher when MFVI_PROF is set.  Columns per wave:
T0 prologue+slab, T1 prefetch, T2 barrier waits, T3 store, T4 weight chunk regs -> LDS, T5 MFMA, T6 epilogue (consumer) /
vmcnt wait (producer), T7 tail (consumer) / producer-side fold.  Restore the file with `git checkout` afterwards; never commit the patched kernel."""
import sys
p = sys.argv[1] if len(sys.argv) > 1 else 'mfvi-dip-mia_amd/csrc/conv_mfma.hip'
s = open(p).read()
def rep(a, b, n=1):
    global s
    assert a in s, a[:60]
    s = s.replace(a, b, n)
rep("    int nx, ny, nz;", "    int nx, ny, nz; long long* prof;")
rep("    using Cfg = MCfg<KS, STRIDE, MF, TH, BIGC, REM>;\n    constexpr int TW = Cfg::TW, CT = Cfg::CT,", "    long long T[8] = {0,0,0,0,0,0,0,0}; long long tc = clock64();\n#define TICK(i) { const long long n_ = clock64(); T[i] += n_ - tc; tc = n_; }\n    using Cfg = MCfg<KS, STRIDE, MF, TH, BIGC, REM>;\n    constexpr int TW = Cfg::TW, CT = Cfg::CT,")
rep("    const int H = g.H, W = g.W;\n    const int SH = MODE == 0 ? H : g.Ho", "    TICK(0)\n    const int H = g.H, W = g.W;\n    const int SH = MODE == 0 ? H : g.Ho")
rep('''        set_tile(ptile); prefetch(pc0); wfetch(pc0);
        __syncthreads();                                  // (S0) channel constants / bias / WS slab visible
        store(pc0, s_x[0]);
        wstore(pc0, s_w);''', '''        set_tile(ptile); prefetch(pc0); wfetch(pc0);
        TICK(1)
        __syncthreads();                                  // (S0) channel constants / bias / WS slab visible
        TICK(2)
        store(pc0, s_x[0]);
        TICK(3)
        wstore(pc0, s_w);
        TICK(4)''')
rep('''        lds_barrier();                                    // (A) chunk 0 published
        for (int it = 0; it < n_iters; ++it) {''', '''        TICK(1)
        lds_barrier();                                    // (A) chunk 0 published
        TICK(2)
        for (int it = 0; it < n_iters; ++it) {''')
rep('''            if (fold_now) { if (fci == 0) fold_fetch(ftile, p0); else if (fci == 1) fold_fetch(ftile, p1); else fold_fetch(ftile, p2); }
            if (it + 1 < n_iters) {
                int nt, nc; chunk_of(it + 1, nt, nc);
                store(nc, s_x[(it + 1) & 1]);
                wstore(nc, s_w + ((it + 1) & 1) * WCHUNK);
                if (it + 2 < n_iters) { int n2, c2; chunk_of(it + 2, n2, c2); if (n2 != ptile) { set_tile(n2); ptile = n2; } prefetch(c2); wfetch(c2); }
            }
            if (fold_now) { if (fci == 0) fold_do(ftile, p0); else if (fci == 1) fold_do(ftile, p1); else fold_do(ftile, p2); }
            lds_barrier();
        }''', '''            if (fold_now) { if (fci == 0) fold_fetch(ftile, p0); else if (fci == 1) fold_fetch(ftile, p1); else fold_fetch(ftile, p2); }
            TICK(7)
            if (it + 1 < n_iters) {
                int nt, nc; chunk_of(it + 1, nt, nc);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TICK(6)
                store(nc, s_x[(it + 1) & 1]);
                TICK(3)
                wstore(nc, s_w + ((it + 1) & 1) * WCHUNK);
                TICK(4)
                if (it + 2 < n_iters) { int n2, c2; chunk_of(it + 2, n2, c2); if (n2 != ptile) { set_tile(n2); ptile = n2; } prefetch(c2); wfetch(c2); }
                TICK(1)
            }
            if (fold_now) { if (fci == 0) fold_do(ftile, p0); else if (fci == 1) fold_do(ftile, p1); else fold_do(ftile, p2); }
            TICK(7)
            lds_barrier();
            TICK(2)
        }''')
rep('''        __syncthreads();                                  // (S0)
        lds_barrier();                                    // (A)
        for (int it = 0; it < n_iters; ++it) {
            const int tile = tile_begin + it / n_chunks''', '''        __syncthreads();                                  // (S0)
        lds_barrier();                                    // (A)
        TICK(2)
        for (int it = 0; it < n_iters; ++it) {
            const int tile = tile_begin + it / n_chunks''')
rep("            if (ci == n_chunks - 1) {\n              if constexpr (PEPI) {", "            TICK(5)\n            if (ci == n_chunks - 1) {\n              if constexpr (PEPI) {")
rep('''            lds_barrier();
        }
        if (do_stats) {''', '''            TICK(6)
            lds_barrier();
            TICK(2)
        }
        if (do_stats) {''')
rep('''                              s_red[0][q][which] + s_red[1][q][which] + s_red[2][q][which] + s_red[3][q][which]);
            }
        }
    }
}''', '''                              s_red[0][q][which] + s_red[1][q][which] + s_red[2][q][which] + s_red[3][q][which]);
            }
        }
    }
    TICK(7)
    if (A.prof && bx == 1 && by == 0 && k == 3 && (threadIdx.x & 63) == 0) for (int i = 0; i < 8; ++i) A.prof[(threadIdx.x >> 6) * 8 + i] = T[i];
}''')
rep("    A.vec_out = MODE == 0 &&",
    "    static long long* prof = [] { long long* p = nullptr; if (getenv(\"MFVI_PROF\")) { (void)hipMalloc((void**)&p, 64 * 8); (void)hipMemset(p, 0, 64 * 8); } return p; }();\n    A.prof = prof;\n    A.vec_out = MODE == 0 &&")
rep("        return (int)hipGetLastError();                                                                                     \\\n    }\n#define GO(MF_, TH_)",
    "        if (prof) { long long h[64]; (void)hipStreamSynchronize(st); (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost); fprintf(stderr, \"MODE %d KS %d mf %d th %d flat %d ff %d T %d tiles %d my %d chunks/tile %d\\n\", MODE, KS, MF_, TH_, (int)(FL_), (int)ff, A.tiles_per_block, A.n_tiles, my, (RED + Cfg::CC - 1) / Cfg::CC); for (int w = 0; w < 8; ++w) { fprintf(stderr, \"wave %d:\", w); for (int i = 0; i < 8; ++i) fprintf(stderr, \" %lld\", h[w * 8 + i]); fprintf(stderr, \"\\n\"); } } \\\n        return (int)hipGetLastError();                                                                                     \\\n    }\n#define GO(MF_, TH_)")
rep("#include <cstdlib>", "#include <cstdlib>\n#include <cstdio>")
open(p, 'w').write(s)
