// Placement probe (DESIGN §5 "Placement"): where in HBM do the output arrays of the march have to sit?
//
// Physical memory is taken in 1-GiB handles through the HIP virtual-memory API (hipMemCreate), each
// mapped at its own virtual address, so that every experiment below addresses *chosen physical chunks*
// instead of whatever a malloc returned.  The writer kernel has the store structure of k_trace_iso's
// path mode (72 row streams, one 16-B store per lane per row, 128-thread blocks) and no arithmetic.
//
//   E1  pair matrix: 36 rows in chunk i + 36 rows in chunk j, for every pair of a chunk subset
//   E2  all chunks against a few reference chunks (the "kind" of every chunk, 1-GiB resolution)
//   E3  row-to-chunk assignments: blocked (36|36), alternating, spread over 3 / 4 / 8 / 72 chunks
//   E4  arrays striped over two kinds in 2-MiB pieces (separate 2-MiB handles; the mapping is verified
//       by data: every piece is also mapped at an address of its own and tagged there)
//   E5  read streams and copies: does the kind matter for loads?
//   `census` mode (argv[1] = "census"): takes (nearly) all of the HBM in 1-GiB chunks and prints the kind
//       of every chunk -- how much memory there is of each kind
//
// Build:  hipcc -O2 --offload-arch=gfx950 -o benchmarks/vmm_placement_probe benchmarks/vmm_placement_probe.hip
// Output: plain text / CSV on stdout (commit what matters under profiles/).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); fflush(stdout); exit(2); } } while (0)

#define MAXROWS 72
struct Rows { double* p[MAXROWS]; };

__global__ __launch_bounds__(128) void k_write(Rows rows, int nrows, int64_t n, double v) {
    const int64_t i = ((int64_t)blockIdx.x * 128 + threadIdx.x) * 2;
    if (i >= n) return;
    d2 val = {v + (double)i, v - (double)i};
    for (int r = 0; r < nrows; ++r) { *(d2*)(rows.p[r] + i) = val; val += 1.0; }
}

__global__ __launch_bounds__(256) void k_fill(double* p, int64_t n, double v) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; i < n; i += (int64_t)gridDim.x * 512) {
        d2 val = {v, v}; *(d2*)(p + i) = val;
    }
}

// 72 read streams: every lane loads 16 B from each row; the sum keeps the loads alive
__global__ __launch_bounds__(128) void k_read(Rows rows, int nrows, int64_t n, double* sink) {
    const int64_t i = ((int64_t)blockIdx.x * 128 + threadIdx.x) * 2;
    if (i >= n) return;
    d2 acc = {0.0, 0.0};
    for (int r = 0; r < nrows; ++r) acc += *(const d2*)(rows.p[r] + i);
    if (acc.x + acc.y == 12345.678) sink[0] = acc.x;
}

// the traffic of the path-mode march without its arithmetic: every lane loads 16 B from each of the
// `nin` input rows (x0, k0, E0: 9 rows), then stores 16 B to each of the 72 output rows
struct InRows { const double* p[9]; };
__global__ __launch_bounds__(128) void k_march_traffic(InRows in, int nin, Rows rows, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 128 + threadIdx.x) * 2;
    if (i >= n) return;
    d2 val = {0.0, 0.0};
    for (int r = 0; r < nin; ++r) val += *(const d2*)(in.p[r] + i);
    for (int r = 0; r < MAXROWS; ++r) { *(d2*)(rows.p[r] + i) = val; val += 1.0; }
}

// rows [0, nrows/2) are read, rows [nrows/2, nrows) written (36 load + 36 store streams)
__global__ __launch_bounds__(128) void k_copy(Rows rows, int nrows, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 128 + threadIdx.x) * 2;
    if (i >= n) return;
    const int h = nrows / 2;
    for (int r = 0; r < h; ++r) *(d2*)(rows.p[h + r] + i) = *(const d2*)(rows.p[r] + i);
}

__global__ void k_tag(double* p, double v) { p[0] = v; }
__global__ void k_peek(const double* p, double* out, int slot) { out[slot] = p[0]; }

static hipEvent_t ev_a, ev_b;

template <typename F>
static float time_launch(F launch, int warm, int reps) {
    for (int i = 0; i < warm; ++i) launch();
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(ev_a, 0));
        launch();
        CHECK(hipEventRecord(ev_b, 0));
        CHECK(hipEventSynchronize(ev_b));
        float ms; CHECK(hipEventElapsedTime(&ms, ev_a, ev_b));
        best = std::min(best, ms);
    }
    return best;
}

static float time_rows(const Rows& rows, int nrows, int64_t n, int warm, int reps) {
    dim3 grid((unsigned)((n / 2 + 127) / 128));
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(k_write, grid, dim3(128), 0, 0, rows, nrows, n, 1.0);
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(ev_a, 0));
        hipLaunchKernelGGL(k_write, grid, dim3(128), 0, 0, rows, nrows, n, 1.0);
        CHECK(hipEventRecord(ev_b, 0));
        CHECK(hipEventSynchronize(ev_b));
        float ms; CHECK(hipEventElapsedTime(&ms, ev_a, ev_b));
        best = std::min(best, ms);
    }
    return best;
}

int main(int argc, char** argv) {
    const bool census = argc > 1 && strcmp(argv[1], "census") == 0;
    int64_t nch_req = census ? 100000 : (argc > 1 ? atoll(argv[1]) : 224);     // chunks of 1 GiB
    int sub = argc > 2 ? atoi(argv[2]) : 2;                // E1 uses every sub-th chunk
    CHECK(hipSetDevice(0));
    CHECK(hipEventCreate(&ev_a)); CHECK(hipEventCreate(&ev_b));
    size_t free_b, total_b; CHECK(hipMemGetInfo(&free_b, &total_b));
    printf("# hipMemGetInfo free %.2f GiB total %.2f GiB\n", free_b / 1073741824.0, total_b / 1073741824.0);

    hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran_min = 0, gran_rec = 0;
    CHECK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
    CHECK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
    printf("# VMM granularity min %zu recommended %zu\n", gran_min, gran_rec);

    const size_t CH = (size_t)1 << 30;
    int64_t nch = std::min<int64_t>(nch_req, (int64_t)(free_b / CH) - (census ? 3 : 24));
    if (nch < 8) { printf("not enough memory\n"); return 1; }
    std::vector<hipMemGenericAllocationHandle_t> h(nch);
    for (int64_t i = 0; i < nch; ++i) CHECK(hipMemCreate(&h[i], CH, &prop, 0));
    void* va = nullptr;
    CHECK(hipMemAddressReserve(&va, CH * nch, CH, nullptr, 0));
    hipMemAccessDesc acc; memset(&acc, 0, sizeof acc);
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int64_t i = 0; i < nch; ++i) CHECK(hipMemMap((char*)va + i * CH, CH, 0, h[i], 0));
    CHECK(hipMemSetAccess(va, CH * nch, &acc, 1));
    printf("# mapped %lld chunks of 1 GiB at %p\n", (long long)nch, va);
    auto chunk = [&](int64_t i) { return (double*)((char*)va + i * CH); };

    // first touch + per-chunk fill rate (sample)
    for (int64_t i = 0; i < nch; ++i) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, chunk(i), (int64_t)(CH / 8), 0.0);
    CHECK(hipDeviceSynchronize());
    printf("# E0 single-stream fill of one chunk, TB/s:");
    for (int64_t i = 0; i < nch; i += std::max<int64_t>(1, nch / 16)) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CHECK(hipEventRecord(ev_a, 0));
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, chunk(i), (int64_t)(CH / 8), 1.0);
            CHECK(hipEventRecord(ev_b, 0)); CHECK(hipEventSynchronize(ev_b));
            float ms; CHECK(hipEventElapsedTime(&ms, ev_a, ev_b)); best = std::min(best, ms);
        }
        printf(" [%lld] %.2f", (long long)i, CH / 1e9 / best);
    }
    printf("\n");

    // 36 rows per chunk: row length in doubles, multiple of 512
    const int64_t L = (int64_t)(CH / 36 / 8) / 512 * 512;
    const double gb = 72.0 * L * 8 / 1e9;
    auto pair_rows = [&](int64_t i, int64_t j) {
        Rows r;
        for (int k = 0; k < 36; ++k) r.p[k] = chunk(i) + (int64_t)k * L;
        for (int k = 0; k < 36; ++k) r.p[36 + k] = chunk(j) + (int64_t)k * L;
        return r;
    };
    printf("# rows of %lld doubles, %.3f GB per launch\n", (long long)L, gb);

    if (census) {
        // kind of every chunk: the representative it is slow with (representatives found on the way)
        std::vector<int64_t> rep;
        std::vector<int> kind(nch, -1);
        std::vector<int64_t> count;
        for (int64_t j = 0; j < nch; ++j) {
            int found = -1;
            for (size_t q = 0; q < rep.size() && found < 0; ++q)
                if (gb / time_rows(pair_rows(rep[q], j), 72, L, 1, 3) < 6.3) found = (int)q;
            if (found < 0) { found = (int)rep.size(); rep.push_back(j); count.push_back(0); }
            kind[j] = found; count[found] += 1;
        }
        printf("CENSUS %lld chunks of 1 GiB in allocation order, kind of each:\n", (long long)nch);
        for (int64_t j = 0; j < nch; ++j) { putchar('A' + kind[j]); if (j % 96 == 95) putchar('\n'); }
        printf("\n");
        for (size_t q = 0; q < rep.size(); ++q)
            printf("kind %c: %lld GiB (first chunk %lld)\n", (char)('A' + q), (long long)count[q], (long long)rep[q]);
        printf("done\n");
        return 0;
    }

    // E1: pair matrix over a subset
    std::vector<int64_t> S;
    for (int64_t i = 0; i < nch; i += sub) S.push_back(i);
    const int m = (int)S.size();
    std::vector<float> M((size_t)m * m, 0.f);
    for (int a = 0; a < m; ++a)
        for (int b = a + 1; b < m; ++b) {
            float t = time_rows(pair_rows(S[a], S[b]), 72, L, 1, 3);
            M[(size_t)a * m + b] = M[(size_t)b * m + a] = t;
        }
    printf("E1 pair matrix, TB/s x100 (rows/cols = chunk index, step %d)\n", sub);
    printf("     ");
    for (int b = 0; b < m; ++b) printf("%4lld", (long long)S[b]);
    printf("\n");
    for (int a = 0; a < m; ++a) {
        printf("%4lld:", (long long)S[a]);
        for (int b = 0; b < m; ++b) {
            if (a == b) printf("   ."); else printf("%4d", (int)(gb / M[(size_t)a * m + b] * 100 + 0.5));
        }
        printf("\n");
    }
    fflush(stdout);

    // E2: every chunk against reference chunks
    int64_t refs[4] = {0, nch / 3, 2 * nch / 3, nch - 1};
    printf("E2 all chunks vs reference chunks, TB/s\nchunk");
    for (int r = 0; r < 4; ++r) printf(",ref%lld", (long long)refs[r]);
    printf("\n");
    for (int64_t j = 0; j < nch; ++j) {
        printf("%lld", (long long)j);
        for (int r = 0; r < 4; ++r) {
            if (refs[r] == j) { printf(","); continue; }
            printf(",%.3f", gb / time_rows(pair_rows(refs[r], j), 72, L, 1, 3));
        }
        printf("\n");
    }
    fflush(stdout);

    // E3: row -> chunk assignments.  Chunks given as a list; row r -> list[r % len], slot r / len.
    auto assign_rows = [&](const std::vector<int64_t>& list, bool blocked) {
        Rows r; const int len = (int)list.size();
        std::vector<int> used(nch, 0);
        for (int k = 0; k < 72; ++k) {
            int64_t c = blocked ? list[(size_t)k * len / 72] : list[k % len];
            r.p[k] = chunk(c) + (int64_t)(used[c]++) * L;
        }
        return r;
    };
    // find best and worst partner of chunk 0 from E1 row 0
    int bestb = 1, worstb = 1;
    for (int b = 1; b < m; ++b) {
        if (M[b] < M[bestb]) bestb = b;
        if (M[b] > M[worstb]) worstb = b;
    }
    const int64_t cA = S[0], cFast = S[bestb], cSlow = S[worstb];
    printf("E3 assignments (chunk A=%lld, fast partner=%lld, slow partner=%lld)\n",
           (long long)cA, (long long)cFast, (long long)cSlow);
    struct Case { const char* name; std::vector<int64_t> list; bool blocked; };
    std::vector<Case> cases;
    cases.push_back({"blocked  A|slow", {cA, cSlow}, true});
    cases.push_back({"alternate A,slow", {cA, cSlow}, false});
    cases.push_back({"blocked  A|fast", {cA, cFast}, true});
    cases.push_back({"alternate A,fast", {cA, cFast}, false});
    { std::vector<int64_t> l; for (int k = 0; k < 4; ++k) l.push_back(k * (nch - 1) / 3); cases.push_back({"4 chunks spread, blocked", l, true}); cases.push_back({"4 chunks spread, alternating", l, false}); }
    { std::vector<int64_t> l; for (int k = 0; k < 8; ++k) l.push_back(k * (nch - 1) / 7); cases.push_back({"8 chunks spread, blocked", l, true}); cases.push_back({"8 chunks spread, alternating", l, false}); }
    { std::vector<int64_t> l; for (int k = 0; k < 8; ++k) l.push_back(k); cases.push_back({"8 adjacent chunks, blocked", l, true}); cases.push_back({"8 adjacent chunks, alternating", l, false}); }
    { std::vector<int64_t> l; for (int k = 0; k < 72; ++k) l.push_back(k * (nch - 1) / 71); cases.push_back({"72 chunks spread (one row each)", l, false}); }
    { std::vector<int64_t> l; for (int k = 0; k < 72; ++k) l.push_back(k); cases.push_back({"72 adjacent chunks (one row each)", l, false}); }
    for (auto& c : cases) {
        float t = time_rows(assign_rows(c.list, c.blocked), 72, L, 2, 5);
        printf("  %-36s %.4f ms  %.3f TB/s\n", c.name, t, gb / t);
    }
    fflush(stdout);

    // E3b: full-size arrays (9 994 752-double rows = the bench's pitch): x_hit 36 rows, k_out 36 rows,
    // each array = 3 consecutive chunks (virtually contiguous, 2.88 GB), at chosen chunk offsets.
    {
        const int64_t N = 9994476, P = 9994752;
        const double gbf = 72.0 * N * 8 / 1e9;
        printf("E3b full-size arrays (36 rows x %lld doubles each), x at chunk 0.., k at chunk c..\n", (long long)P);
        auto full_rows = [&](int64_t cx, int64_t ck) {
            Rows r;
            for (int k = 0; k < 36; ++k) r.p[k] = chunk(cx) + (int64_t)k * P;
            for (int k = 0; k < 36; ++k) r.p[36 + k] = chunk(ck) + (int64_t)k * P;
            return r;
        };
        for (int64_t ck = 3; ck + 3 <= nch; ck += std::max<int64_t>(1, (nch - 6) / 40)) {
            float t = time_rows(full_rows(0, ck), 72, N, 2, 5);
            printf("  x@0 k@%-4lld %.4f ms %.3f TB/s\n", (long long)ck, t, gbf / t);
        }
    }
    fflush(stdout);

    // E5: loads.  72 read streams (36 | 36) and 36-read + 36-write copies, same kind vs two kinds.
    {
        double* sink; CHECK(hipMalloc(&sink, 64));
        const dim3 grid((unsigned)((L / 2 + 127) / 128));
        struct { const char* name; int64_t a, b; } pairs[] = {{"same kind ", cA, cSlow}, {"two kinds ", cA, cFast}};
        for (auto& q : pairs) {
            Rows r = pair_rows(q.a, q.b);
            float tr = time_launch([&] { hipLaunchKernelGGL(k_read, grid, dim3(128), 0, 0, r, 72, L, sink); }, 2, 5);
            float tc = time_launch([&] { hipLaunchKernelGGL(k_copy, grid, dim3(128), 0, 0, r, 72, L); }, 2, 5);
            float tw = time_rows(r, 72, L, 2, 5);
            printf("E5 %s (%lld,%lld): 72 read streams %.3f TB/s   36 read -> 36 write %.3f TB/s   72 write streams %.3f TB/s\n",
                   q.name, (long long)q.a, (long long)q.b, gb / tr, gb / tc, gb / tw);
        }
        CHECK(hipFree(sink));
    }
    fflush(stdout);

    // E6: the march's traffic mix (9 load streams = 11 % of the bytes + 72 store streams), no arithmetic:
    // the structural floor of k_trace_iso's path mode, by where the inputs live
    {
        const dim3 grid((unsigned)((L / 2 + 127) / 128));
        // a chunk of a third kind: fast with both A and its fast partner
        int64_t cThird = -1;
        for (int b = 1; b < m && cThird < 0; ++b)
            if (S[b] != cFast && S[b] != cSlow && M[b] < 0.9 * M[worstb] && M[(size_t)bestb * m + b] < 0.9 * M[worstb])
                cThird = S[b];
        struct { const char* name; int64_t out_a, out_b, in_c; } cases6[] = {
            {"outputs same kind, inputs in that kind   ", cA, cSlow, cSlow + 1 < nch ? cSlow + 1 : cSlow},
            {"outputs two kinds, inputs with x_hit     ", cA, cFast, cA + 1},
            {"outputs two kinds, inputs in a third kind", cA, cFast, cThird}};
        for (auto& q : cases6) {
            if (q.in_c < 0) { printf("E6 %s: no third kind among the sampled chunks\n", q.name); continue; }
            Rows r = pair_rows(q.out_a, q.out_b);
            InRows in;
            for (int k = 0; k < 9; ++k) in.p[k] = chunk(q.in_c) + (int64_t)k * L;      // (9 rows fit: 36 per chunk)
            float t0 = time_launch([&] { hipLaunchKernelGGL(k_march_traffic, grid, dim3(128), 0, 0, in, 0, r, L); }, 2, 5);
            float t9 = time_launch([&] { hipLaunchKernelGGL(k_march_traffic, grid, dim3(128), 0, 0, in, 9, r, L); }, 2, 5);
            printf("E6 %s (out %lld|%lld, in %lld): stores only %.3f TB/s; 9 load + 72 store streams %.3f TB/s (%.4f ms)\n",
                   q.name, (long long)q.out_a, (long long)q.out_b, (long long)q.in_c, gb / t0,
                   (72.0 + 9.0) * L * 8 / 1e9 / t9, t9);
        }
    }
    fflush(stdout);

    // E4: arrays striped over two kinds in 2-MiB pieces.  One chunk of each kind is given back to the
    // driver and taken again as 512 handles of 2 MiB.  Every piece is mapped at an address of its own
    // (where it is tagged) and inside a "virtual chunk" -- layout 0: each virtual chunk is built from the
    // pieces of its own source; layout 1: pieces alternate between the two sources, so both virtual
    // chunks are half/half mixes.  The tags read back through the virtual chunks prove what is mapped
    // where.  `fresh` = every layout gets newly reserved addresses; `reused` = layout 1 is mapped over
    // the addresses layout 0 had (after hipMemUnmap) -- the case in which this ROCm stack keeps
    // translating to the OLD pages.
    {
        const size_t PIECE = (size_t)2 << 20; const int NP = (int)(CH / PIECE);
        int64_t src[2] = {cA + 1 < nch ? cA + 1 : cA, cFast};
        std::vector<hipMemGenericAllocationHandle_t> pieces[2];
        void* own_va[2];
        double* peek; CHECK(hipMalloc(&peek, 8 * sizeof(double)));
        for (int s = 0; s < 2; ++s) {
            CHECK(hipMemUnmap((char*)va + src[s] * CH, CH));
            CHECK(hipMemRelease(h[src[s]]));
            pieces[s].resize(NP);
            for (int k = 0; k < NP; ++k) CHECK(hipMemCreate(&pieces[s][k], PIECE, &prop, 0));
            CHECK(hipMemAddressReserve(&own_va[s], CH, PIECE, nullptr, 0));
            for (int k = 0; k < NP; ++k) CHECK(hipMemMap((char*)own_va[s] + (size_t)k * PIECE, PIECE, 0, pieces[s][k], 0));
            CHECK(hipMemSetAccess(own_va[s], CH, &acc, 1));
        }
        void* vc[2] = {nullptr, nullptr};      // the two virtual chunks
        struct { const char* name; int layout; bool fresh; } runs[] = {
            {"layout 0 (own pieces), fresh addresses", 0, true},
            {"layout 1 (alternating pieces), fresh addresses", 1, true},
            {"layout 0 (own pieces), fresh addresses", 0, true},
            {"layout 1 (alternating pieces), mapped over the addresses of the run before (after hipMemUnmap)", 1, false}};
        for (auto& run : runs) {
            for (int s = 0; s < 2; ++s) {
                if (vc[s]) CHECK(hipMemUnmap(vc[s], CH));
                if (run.fresh) CHECK(hipMemAddressReserve(&vc[s], CH, PIECE, nullptr, 0));      // old range: left alone
                for (int k = 0; k < NP; ++k) {
                    const int from = run.layout == 0 ? s : (k + s) % 2;
                    CHECK(hipMemMap((char*)vc[s] + (size_t)k * PIECE, PIECE, 0, pieces[from][k], 0));
                }
                CHECK(hipMemSetAccess(vc[s], CH, &acc, 1));
            }
            double* c0 = (double*)vc[0]; double* c1 = (double*)vc[1];
            for (int s = 0; s < 2; ++s) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (double*)vc[s], (int64_t)(CH / 8), 0.0);
            for (int s = 0; s < 2; ++s)
                for (int k = 0; k < 4; ++k)
                    hipLaunchKernelGGL(k_tag, dim3(1), dim3(1), 0, 0, (double*)((char*)own_va[s] + (size_t)k * PIECE), 100.0 * s + k + 1000.0);
            for (int s = 0; s < 2; ++s)
                for (int k = 0; k < 4; ++k)
                    hipLaunchKernelGGL(k_peek, dim3(1), dim3(1), 0, 0, (const double*)((char*)vc[s] + (size_t)k * PIECE), peek, 4 * s + k);
            double hp[8]; CHECK(hipMemcpy(hp, peek, sizeof hp, hipMemcpyDeviceToHost));
            printf("E4 %s\n   tags (1000 + 100*source + piece) seen in the first 4 pieces of virtual chunk 0: %g %g %g %g   of 1: %g %g %g %g\n",
                   run.name, hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7]);
            auto two = [&](double* lo, double* hi) {
                Rows r;
                for (int k = 0; k < 36; ++k) r.p[k] = lo + (int64_t)k * L;
                for (int k = 0; k < 36; ++k) r.p[36 + k] = hi + (int64_t)k * L;
                return r;
            };
            printf("   virtual chunk 0 | virtual chunk 1: %.3f TB/s\n", gb / time_rows(two(c0, c1), 72, L, 2, 5));
            for (int s = 0; s < 2; ++s)
                for (int r = 0; r < 4; ++r) {
                    if (refs[r] == src[0] || refs[r] == src[1]) continue;
                    printf("   reference chunk %lld | virtual chunk %d: %.3f TB/s\n", (long long)refs[r], s,
                           gb / time_rows(two(chunk(refs[r]), (double*)vc[s]), 72, L, 2, 5));
                }
            {
                Rows r; const int64_t L2 = L / 2 / 512 * 512;
                for (int k = 0; k < 72; ++k) r.p[k] = c0 + (int64_t)k * L2;
                printf("   all 72 rows inside virtual chunk 0: %.3f TB/s\n", 72.0 * L2 * 8 / 1e9 / time_rows(r, 72, L2, 2, 5));
            }
            CHECK(hipDeviceSynchronize());
        }
    }
    printf("done\n");
    return 0;
}
