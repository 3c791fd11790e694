// prt_kernels.h -- the __global__ kernels of libprt.so (launched from prt.hip).
//
//   k_trace_iso<MODE,VEC_IN,VEC_OUT,SHAPES,MOMENTS,UNI,IMG>   whole isotropic sequence in one launch
//   k_trace_general<MODE>                              whole sequence through crystals (leaf re-tracing)
//   k_propagate / k_interact_iso / k_interact_aniso    one plugin-granular step (tight arrays, one ray per thread)
//   k_propagate_rows / k_interact_iso_rows             the same for big isotropic bundles (row-pitched, two rays per thread)
//   k_shape_eval, k_efield_perp, k_poynting_dir, k_path_sums
//   k_moments_partial / k_moments_final                deterministic bundle moments
//   k_compact_count / k_compact_scan / k_compact_scatter   order-preserving compaction
//   k_rectgrid_mask / k_rectgrid_scatter               RectGrid raster + collimated bundle
//   k_raster_mask / k_raster_bundle                    any outer-product raster + collimated / divergent bundle
#pragma once
#include "prt_device.h"
#include "prt_aniso.h"
#include "prt_aniso_cplx.h"

#define PRT_BLOCK 256
#define CMP_ITEMS 4  // mask bytes per thread in the compaction / raster kernels
#define CMP_TILE (PRT_BLOCK * CMP_ITEMS)

// ---------------------------------------------------------------------------
// ray load / store helpers for the fused march.  Arrays are component-major rows
// with a row PITCH (in elements): element (row r, ray i) lives at r*pitch + i.
// A thread owns the two adjacent rays i, i+1.  With an even pitch and a 16-byte
// aligned base every row is 16-B aligned and one dwordx4 access moves both rays
// (VEC = true); otherwise two 8-B accesses with a tail guard.
//
// Row alignment decides the achievable HBM WRITE bandwidth: with rows that do not
// start on a 128-B line a wave's 1-KiB store is split over partial lines and the
// 72-stream path write drops from ~6.3 to ~3.9-4.7 TB/s (measured, DESIGN.md
// "row pitch").  prt_recommended_pitch() rounds the pitch to 4 KiB.
// ---------------------------------------------------------------------------
typedef double prt_double2 __attribute__((ext_vector_type(2)));

template <bool VEC>
struct rayio {
    // second = false: ray i+1 does not exist (odd N tail) -> duplicate ray i
    static PRT_DEV void load(const double *__restrict__ a, int64_t pitch, int64_t i, bool second,
                             vec3 v[2]) {
        if (VEC) {
            // non-temporal hint: every input is read once (+1.7 % on the path-mode march, +4 % in image
            // mode, same arrays: profiles/r02*_ab_variants*.json)
            const prt_double2 x = __builtin_nontemporal_load(reinterpret_cast<const prt_double2 *>(a + i));
            const prt_double2 y = __builtin_nontemporal_load(reinterpret_cast<const prt_double2 *>(a + pitch + i));
            const prt_double2 z = __builtin_nontemporal_load(reinterpret_cast<const prt_double2 *>(a + 2 * pitch + i));
            v[0] = v3(x.x, y.x, z.x);
            v[1] = v3(x.y, y.y, z.y);
        } else {
            v[0] = v3(a[i], a[pitch + i], a[2 * pitch + i]);
            v[1] = second ? v3(a[i + 1], a[pitch + i + 1], a[2 * pitch + i + 1]) : v[0];
        }
    }
    static PRT_DEV void store(double *__restrict__ a, int64_t pitch, int64_t i, bool second,
                              const vec3 v[2]) {
        if (VEC) {
            // non-temporal hint: a path array is written once and not read by this kernel again; with
            // x_hit and k_out in two kinds of HBM it is worth 2 % (1.016 vs 1.038 ms, same arrays).  Of the cache-policy bits a
            // gfx950 store can carry, "nt" alone is the best: nt 0.975 ms, sc0 nt 0.976, sc1 nt / sc0 sc1 nt 0.979,
            // none / sc0 sc1 0.996-0.999 (inline-asm stores, profiles/r02zd_ab_store_cache_policy_bits.json)
            __builtin_nontemporal_store(prt_double2{v[0].x, v[1].x}, reinterpret_cast<prt_double2 *>(a + i));
            __builtin_nontemporal_store(prt_double2{v[0].y, v[1].y}, reinterpret_cast<prt_double2 *>(a + pitch + i));
            __builtin_nontemporal_store(prt_double2{v[0].z, v[1].z}, reinterpret_cast<prt_double2 *>(a + 2 * pitch + i));
        } else {
            a[i] = v[0].x;
            a[pitch + i] = v[0].y;
            a[2 * pitch + i] = v[0].z;
            if (second) {
                a[i + 1] = v[1].x;
                a[pitch + i + 1] = v[1].y;
                a[2 * pitch + i + 1] = v[1].z;
            }
        }
    }
    // the masks of a record packed into one byte per ray: bit 0 = lo, bit 1 = hi, bit 2 = nc (PRT_MODE_FLAGS)
    static PRT_DEV void store_flags(uint8_t *__restrict__ m, int64_t i, bool second, const bool lo[2],
                                    const bool hi[2], const bool nc[2]) {
        const unsigned f0 = (lo[0] ? 1u : 0u) | (hi[0] ? 2u : 0u) | (nc[0] ? 4u : 0u);
        const unsigned f1 = (lo[1] ? 1u : 0u) | (hi[1] ? 2u : 0u) | (nc[1] ? 4u : 0u);
        if (VEC) {
            *reinterpret_cast<uint16_t *>(m + i) = (uint16_t)(f0 | (f1 << 8));
        } else {
            m[i] = (uint8_t)f0;
            if (second) m[i + 1] = (uint8_t)f1;
        }
    }
    static PRT_DEV void store_mask(uint8_t *__restrict__ m, int64_t i, bool second, const bool b[2]) {
        if (VEC) {
            *reinterpret_cast<uint16_t *>(m + i) = (uint16_t)((b[0] ? 1u : 0u) | (b[1] ? 0x100u : 0u));
        } else {
            m[i] = b[0] ? 1 : 0;
            if (second) m[i + 1] = b[1] ? 1 : 0;
        }
    }
};

// first-segment direction selector
//   e_mode 0: d = k/|k|        1: E = (0,1,0) (ray.py:71-73)     2: E given (re [, im])
//          3: e_re holds the unit directions themselves (prt_trace_seq's d0)
template <bool VEC>
PRT_DEV void first_direction(int e_mode, const double *__restrict__ e_re,
                             const double *__restrict__ e_im, int64_t pitch, int64_t i, bool second,
                             const vec3 k[2], vec3 d[2]) {
    if (e_mode == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) d[r] = normalized(k[r]);
    } else if (e_mode == 1) {
#pragma unroll
        for (int r = 0; r < 2; ++r) d[r] = poynting_dir(k[r], v3(0, 1, 0), v3(0, 0, 0));
    } else {
        vec3 er[2], ei[2];
        rayio<VEC>::load(e_re, pitch, i, second, er);
        if (e_im) {
            rayio<VEC>::load(e_im, pitch, i, second, ei);
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) ei[r] = v3(0, 0, 0);
        }
        // (mode 3 as a select, not as a branch of its own: a fourth arm in this prologue changed the
        // schedule of the whole march -- 2.5 % on the asphere config, 5 % in image mode, same arrays)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const vec3 p = poynting_dir(k[r], er[r], ei[r]);
            d[r] = v3(e_mode == 3 ? er[r].x : p.x, e_mode == 3 ? er[r].y : p.y, e_mode == 3 ? er[r].z : p.z);
        }
    }
}

// The UNIFORM first segment (prt_trace_ex, k0 == NULL): the bundles of the reference's analysis layer are
// collimated (analysis/optical_system_analysis.py:83-122) -- one wave vector and one E field for all rays.  The
// values travel as kernel arguments (scalar registers); every lane runs the per-ray arithmetic of
// first_direction on them, so the results equal the array form bit for bit, and only x0 is loaded.
struct first_uniform {
    double k[3], er[3], ei[3];
};
PRT_DEV vec3 uniform_first_direction(int e_mode, const first_uniform &fu, const vec3 &k) {
    const vec3 er = v3(fu.er[0], fu.er[1], fu.er[2]);
    if (e_mode == 0) return normalized(k);
    if (e_mode == 1) return poynting_dir(k, v3(0, 1, 0), v3(0, 0, 0));
    if (e_mode == 3) return er;
    return poynting_dir(k, er, v3(fu.ei[0], fu.ei[1], fu.ei[2]));
}

// ---------------------------------------------------------------------------
// fused isotropic march: OpticalElement.seqtrace's loop (optical_element.py:336-375)
// ---------------------------------------------------------------------------
// The surface table is read through the scalar cache (s_load into SGPRs).  The alternative of north_star -- the block
// copies the table into LDS and the march reads the records from there (ds_read broadcast into VGPRs) -- was built and
// measured in rounds 2-5 (template parameter LDS_TAB, removed in round 6): 90 instead of 66 VGPRs, 5 instead of 7 waves,
// +2.5 % in image mode, no difference in path mode.
// SHAPES (prt_device.h): the shape code compiled into the instantiation.  The host guarantees that the
// table holds nothing else: conics only -> no Newton / polynomial code at all (68 VGPRs, 7 waves per SIMD:
// the double Gauss); conics + even aspheres (BASELINE configs[2]); every shape (116 VGPRs, 4 waves).
// MOMENTS = true additionally reduces the image-plane spot moments of the bundle inside the same
// launch: every block writes {count, sum v, sum v*v} (v = last hit point - mref, rays still valid
// after the last interaction) of its 512 rays to moment_partials[blockIdx.x*7..]; k_moments_final
// sums the blocks in a fixed order.  This is RayBundleAnalysis' centroid / RMS spot input
// (analysis/ray_analysis.py:44-86) without a second pass over the image-plane arrays.
#define MOM_VALUES 7
// Loads and stores share one in-order counter on gfx950 (vmcnt).  If an input load can still be pending when
// a loop over surfaces is entered -- on ANY path, e.g. x0 / k0 when the first direction is given -- the compiler
// guards the later uses and overwrites of those registers inside the loop with s_waitcnt vmcnt(n), and from
// the second surface on such a wait is a wait for the wave's own path STORES to be acknowledged by HBM
// (asphere march: a vmcnt(3) after the stores of every surface; crystal march: three full drains per ray
// at the un-park steps).  One explicit wait in front of the loop removes all of them.
// s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt = imm[3:0] | imm[15:14], expcnt = imm[6:4] = 7 and lgkmcnt = imm[11:8] = 15
// left alone)
#define PRT_WAIT_VMEM_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70)
// Image mode of the all-conic march is FP64-VALU bound: 8 waves/SIMD (63 VGPRs, no spills) instead
// of the allocator's 7 is worth 5 % there (0.47 -> 0.446 ms); path mode is HBM bound and unaffected.
// UNI = true: the uniform first segment, decided at compile time (the aligned instantiations); the unaligned
// fall-back instantiations (VEC_IN / VEC_OUT false) decide at run time (uni_rt) instead of doubling their number.
// IMG = true (path mode): the record of the LAST surface goes to separate rows (img_redirect) instead of its
// rows of the path arrays -- a ray-sharded trace deposits its image plane straight into its slot of the
// all-gather's receive buffer.
struct img_redirect {
    double *x, *k;
    uint8_t *valid, *valid_out;
    int64_t pitch;
};
template <int MODE, bool VEC_IN, bool VEC_OUT, int SHAPES = PRT_SHAPES_ALL, bool MOMENTS = false, bool UNI = false,
          bool IMG = false>
// Block size of the fused isotropic march only.  A/B of builds on the SAME arrays in one process
// (scratch/ab_same_buffers.py; placement alone is worth +-10 %): path mode 64 threads 1.038-1.043,
// 128 threads 1.035-1.046, 256 threads 1.055-1.062, 512 threads 1.078-1.080 ms; image mode
// indifferent.  The write-bound march prefers many small blocks (finer-grained refill of the CUs).
#define PRT_MARCH_BLOCK 128
// asphere level: 82 VGPRs on its own = 5 waves; forcing 6 (80 VGPRs, 2-14 spilled dwords) is no faster in
// path mode and 6 % slower in image mode (benchmarks/ab_variants.py asphere)
#define PRT_ASPHERE_WAVES 5
#define PRT_POLY_WAVES 5
#define PRT_NEWTON_IMAGE_WAVES 5      // image mode of the asphere / polynomial levels (round 5 A/B: 5 waves with a 20-B spill against 4 without)
__global__ __launch_bounds__(PRT_MARCH_BLOCK, SHAPES == PRT_SHAPES_CONIC ? (MODE == PRT_MODE_IMAGE ? 8 : 1) : ((SHAPES == PRT_SHAPES_ASPHERE || SHAPES == PRT_SHAPES_POLY) && MODE == PRT_MODE_IMAGE ? PRT_NEWTON_IMAGE_WAVES : (SHAPES == PRT_SHAPES_ASPHERE ? PRT_ASPHERE_WAVES : (SHAPES == PRT_SHAPES_POLY ? PRT_POLY_WAVES : 1)))) void k_trace_iso(
    const prt_dev_surface *__restrict__ tab_g, int32_t S, int64_t N, int64_t in_pitch,
    const double *__restrict__ x0, const double *__restrict__ k0, const double *__restrict__ e_re,
    const double *__restrict__ e_im, int32_t e_mode, int64_t out_pitch,
    double *__restrict__ xh_out, double *__restrict__ k_out, uint8_t *__restrict__ valid_out_hit,
    uint8_t *__restrict__ valid_out_refr, double mref_x = 0.0, double mref_y = 0.0,
    double mref_z = 0.0, double *__restrict__ moment_partials = nullptr, int32_t packed_flags = 0,
    uint8_t *__restrict__ nonconv_out = nullptr, first_uniform fu = first_uniform(), int32_t uni_rt = 0,
    img_redirect img = img_redirect()) {
    const prt_dev_surface *__restrict__ tab = tab_g;
    // Blocks are dealt to the XCDs round robin (block b runs on XCD b % 8: observed, not promised -- speed only).  Every
    // XCD gets ONE contiguous eighth of the bundle, so that its L2 and its translation caches see an eighth of every
    // one of the 84 rows instead of slices all along them: 1e8 rays (rows 0.8 GB apart) 10.35 -> 9.96 ms, nothing
    // lost at 1e7 (round 5, same arrays, builds interleaved: profiles/r05_ab_xcd_swizzle_*.json).  A ray's result does not
    // depend on where it is computed.
    // (the grid is rounded up to a multiple of 8 blocks -- launch_iso_inst --, which makes b -> blk a bijection of
    //  [0, gridDim.x); the up to seven blocks behind the last ray have nothing to do.  A launch whose grid is NOT a
    //  multiple of 8 gets the linear map: the swizzle would skip ray blocks and compute others twice -- ADVICE r5)
    const int64_t per_xcd = (int64_t)gridDim.x / 8;
    const int64_t blk = (gridDim.x & 7u) ? (int64_t)blockIdx.x
                                         : (int64_t)(blockIdx.x % 8) * per_xcd + (int64_t)(blockIdx.x / 8);
    if (blk * (2 * PRT_MARCH_BLOCK) >= N) return;
    const int64_t i = (blk * PRT_MARCH_BLOCK + threadIdx.x) * 2;
    if (!MOMENTS && i >= N) return;
    const bool second = (i + 1 < N);

    vec3 x[2], k[2], d[2];
    bool valid[2] = {true, true};
    if (i < N) {  // (always true without MOMENTS; with MOMENTS the tail threads join the reduction)
    rayio<VEC_IN>::load(x0, in_pitch, i, second, x);
    if (UNI || (!(VEC_IN && VEC_OUT) && uni_rt)) {
        k[0] = k[1] = v3(fu.k[0], fu.k[1], fu.k[2]);
        d[0] = d[1] = uniform_first_direction(e_mode, fu, k[0]);
    } else {
        rayio<VEC_IN>::load(k0, in_pitch, i, second, k);
        first_direction<VEC_IN>(e_mode, e_re, e_im, in_pitch, i, second, k, d);
    }
    double d2 = 1.0;  // |d|^2: unit Poynting direction on the first segment
    // no input load may still be in flight when the march starts: see PRT_WAIT_VMEM_LOADS
    PRT_WAIT_VMEM_LOADS();

    for (int32_t s = 0; s < S; ++s) {
        const prt_dev_surface *__restrict__ sf = tab + s;
        bool vhit[2];
        bool ncv[2];  // Newton hit its iteration cap at this surface (explicit shapes only)
        vec3 nrm[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            vec3 xh, p, g;
            double g2;
            propagate_step<SHAPES>(sf, x[r], d[r], d2, xh, p, g, g2, valid[r], ncv[r]);
            vhit[r] = valid[r];
            nrm[r] = normal_from_grad<SHAPES>(sf, g, g2);
            x[r] = xh;
        }
        // the hit points go out before the interaction is computed: spreading a surface's stores
        // over the iteration is worth 2 % on the write-bound march (1.031 vs 1.052 ms, same arrays,
        // scratch/ab_same_buffers.py)
        // (IMG: the last surface's rows are somewhere else -- wave-uniform pointer / pitch selects, one store sequence)
        const bool redirected = IMG && s == S - 1;
        const int64_t row_pitch = redirected ? img.pitch : out_pitch;
        if (MODE == PRT_MODE_PATH || s == S - 1) {
            const int64_t so = (MODE == PRT_MODE_PATH) ? (int64_t)s : 0;
            rayio<VEC_OUT>::store(redirected ? img.x : xh_out + so * 3 * out_pitch, row_pitch, i, second, x);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            interact_isotropic(sf, nrm[r], k[r], valid[r]);
            // after an isotropic interaction E is perpendicular to k, so the Poynting
            // direction (ray.py:136-152) is parallel to k; the next intersection takes the
            // unnormalised k with |k|^2 = n_after^2 (conic_t / explicit_t are homogeneous in d)
            d[r] = k[r];
        }
        d2 = sf->n_after * sf->n_after;

        if (MODE == PRT_MODE_PATH || s == S - 1) {
            const int64_t so = (MODE == PRT_MODE_PATH) ? (int64_t)s : 0;
            rayio<VEC_OUT>::store(redirected ? img.k : k_out + so * 3 * out_pitch, row_pitch, i, second, k);
            uint8_t *__restrict__ m_hit = redirected ? img.valid : valid_out_hit + so * out_pitch;
            uint8_t *__restrict__ m_refr = redirected ? img.valid_out : (valid_out_refr ? valid_out_refr + so * out_pitch : nullptr);
            if (packed_flags) {
                rayio<VEC_OUT>::store_flags(m_hit, i, second, vhit, valid, ncv);
            } else {
                rayio<VEC_OUT>::store_mask(m_hit, i, second, vhit);
                if (m_refr) rayio<VEC_OUT>::store_mask(m_refr, i, second, valid);
            }
            if (SHAPES != PRT_SHAPES_CONIC && nonconv_out)
                rayio<VEC_OUT>::store_mask(nonconv_out + so * out_pitch, i, second, ncv);
        }
    }
    }  // i < N
    if (MOMENTS) {
        double acc[MOM_VALUES] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (i + r < N && valid[r]) {
                const double vx = x[r].x - mref_x, vy = x[r].y - mref_y, vz = x[r].z - mref_z;
                acc[0] += 1.0;
                acc[1] += vx;
                acc[2] += vy;
                acc[3] += vz;
                acc[4] += vx * vx;
                acc[5] += vy * vy;
                acc[6] += vz * vz;
            }
        }
        __shared__ double sh[PRT_MARCH_BLOCK / 64][MOM_VALUES];
#pragma unroll
        for (int q = 0; q < MOM_VALUES; ++q) {
            double v = acc[q];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][q] = v;
        }
        __syncthreads();
        if (threadIdx.x < MOM_VALUES) {
            double v = 0.0;
            for (int w = 0; w < PRT_MARCH_BLOCK / 64; ++w) v += sh[w][threadIdx.x];
            moment_partials[blk * MOM_VALUES + threadIdx.x] = v;      // row = block of RAYS, whatever XCD computed it
        }
    }
}

// ---------------------------------------------------------------------------
// fused march through tables that contain anisotropic media (ray doubling).  Thread i owns
// input ray i and ALL its descendants: with A anisotropic interfaces there are 2^A leaves,
// leaf L having bit j = which of the two transmitted solutions is followed at the j-th crystal
// interface.  The tree is walked depth first: at a crystal interface the thread writes both
// children, parks child 1 (hit point, wave vector, direction, mask: 10 values in a per-thread slot
// of its level -- at most one parked child per level, so A slots in private memory) and goes on
// with child 0; at the end of the sequence it resumes the deepest parked child.  Every state of
// the tree is computed exactly once: 2^A - 1 interface solves per input ray whatever A is (the
// previous scheme re-traced shared prefixes -- 2^(A-1) passes from the start -- and was limited to
// A <= 4), in one launch, with no intermediate arrays and no direction buffers.  The walk is the
// same for every ray, so control flow (surface index, level, offsets) is wave-uniform.
// Outputs use the concatenated layout of include/prt.h (rays of a split bundle stacked
// [sol2, sol3] like np.hstack, material_anisotropic.py:89) with ray pitch P >= N: at a level with a
// doublings, leaf L sits at i + P (L mod 2^a).
// ---------------------------------------------------------------------------
#define PRT_FUSED_MAX_CRYSTALS 8
// non-temporal hint on the path stores of the crystal march (written once, never read back by the
// kernel): 0.213 instead of 0.254 ms on BASELINE configs[3] (profiles/r02zc_ab_crystal_store_variants.json)
#define PRT_GSTORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
// ... but NOT on the byte masks: a wave's 64 mask bytes are half a cache line, and with the hint each of them
// goes to memory on its own -- 0.164 ms with the hint, 0.149 without, 0.149 with no mask stores at all
// (profiles/r02zc_ab_crystal_store_variants.json)
#define PRT_GSTORE_MASK(ptr, val) (*(ptr) = (val))
// (three waves with 165 VGPRs and no spill are 4-5 % slower: profiles/r05_ab_crystal_general_3_waves_no_spill.json)
#define PRT_GENERAL_WAVES 4
// ... and of the instantiation for conic surfaces + uniaxial / isotropic crystals without E output (BASELINE
// configs[3]), which needs 79 VGPRs and 98 B of LDS per thread
#define PRT_UNIAXIAL_WAVES 4
// What is parked.  Tables whose crystals are all uniaxial / isotropic (GENERAL = false): hit point, the child's wave
// vector in the frame of the crystal, and a byte (alive | extraordinary << 1) -- 6 doubles + 1 byte; k and the ray
// direction are rebuilt from them when the child is taken up (closed_form_ray, prt_aniso.h: bit-identical to what
// the interface computed).  98 B per thread for two levels: six blocks of 256 threads per CU, where the 146 B of
// (x, k, d, alive) allowed four.  Biaxial crystals (GENERAL): hit point, k, d, alive -- 9 doubles + 1 byte.
// PARK_LDS: the parking slots live in LDS ([level][value][thread]: conflict-free) instead of private memory.  For up to PRT_PARK_LDS_LEVELS crystal interfaces the block's
// slots (37 KB) leave room for four blocks per CU.  Private-memory slots are real HBM / L2 traffic: PMC
// 1.20 GB per launch instead of 0.82 GB on BASELINE configs[3] (0.75 GB algorithmic), and 0.193 instead of
// 0.162 ms in path mode, 0.139 instead of 0.125 ms in image mode (same arrays, benchmarks/ab_crystal.py).
#define PRT_PARK_LDS_LEVELS 2
// A pointer every lane of the wave agrees on, pinned to scalar registers: "p + threadIdx" then compiles to
// the scalar-base form of global_load / global_store (64-bit base in SGPRs + 32-bit lane offset) instead of
// a 64-bit vector add per access -- and the optimiser cannot fold the lane index back into the base.
// (the integer round trip loses the address space; the result is declared global again, or the stores would
// be flat_store)
template <typename T>
PRT_DEV PRT_GLOBAL_AS T *uniform_ptr(T *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (PRT_GLOBAL_AS T *)(((uint64_t)hi << 32) | lo);
}

// Block size of the crystal march alone (the walk is the same for every ray, so blocks share nothing but
// the LDS parking slots): PRT_GENERAL_BLOCK threads.
#define PRT_GENERAL_BLOCK 256

// THE WALK PROGRAM.  The depth-first walk through the tree of split rays is the same for every ray of every
// launch on a table: which surface comes next, at which level, where a parked child is resumed, at which
// (pitch-relative) offsets the records land.  prt_system_create writes it down once as a list of steps
// (host: build_walk_program); the kernel executes the list.  What the kernel used to work out per step in
// scalar arithmetic -- the offsets of the concatenated layout, 64-bit multiplies per row base; a scan of the table
// at every resume -- is read instead, and the loads of a step go out together at its top: the two halves of the
// surface's hot block (prt_device.h) and the walk entry of the NEXT step -- one scalar-memory round trip per step.
// (Fetching the next step's hot block a step ahead as well was tried: the 32 extra SGPRs spilled)
//   s         surface of this step
//   a         doublings so far = level of a crystal interface met at this step        (a, resume, lp, last: a
//   resume    j + 1: before this step the child parked at level j is taken up; 0: the state is carried on  byte each)
//   lp        which of the 2^a branches of this level the step's ray is (row block of the concatenated layout)
//   cum_in    sum of 2^a over the earlier surfaces: offset of the surface's block of x_hit / valid, in units of
//             the ray pitch P; the same for k_out / valid_out is cum_in + 2^a - 1
//   s_park    the surface at which the child taken up by `resume` was parked
//   last      1: s is the last surface of the table (the one record image mode writes)
struct walk_step {
    int32_t s;
    int32_t bits;     // a | resume << 8 | lp << 16 | last << 24
    int32_t cum_in;   // (cum_out = cum_in + 2^a - 1: every earlier crystal interface of level j added 2^j)
    int32_t s_park;   // resume: the surface whose crystal interface parked the child that is taken up
};
static_assert(sizeof(walk_step) == 16, "walk entry = one s_load_dwordx4");

typedef double prt_d8 __attribute__((ext_vector_type(8)));
typedef int32_t prt_i4 __attribute__((ext_vector_type(4)));
// volatile: the load stays where it is written (the optimiser would sink a plain load of constant memory into the
// basic block of its first use -- back to one round trip per field) and stays one wide scalar load.
// The hot block comes in two halves (prt_device.h): the EARLY half (what the intersection and the aperture need)
// and the LATE half (what the interaction needs).
PRT_DEV prt_d8 load_hot_half(const prt_hot_surface *hot, int32_t s, int half) {
    const PRT_CONST_AS volatile prt_d8 *p = (const PRT_CONST_AS volatile prt_d8 *)(uint64_t)(hot + s);
    return p[half];
}
PRT_DEV void unpack_hot_early(const prt_d8 &h0, const prt_dev_surface *full, hot_rec &r) {
    const uint64_t w = __builtin_bit_cast(uint64_t, (double)h0[0]);
    const uint32_t bits = (uint32_t)w;
    r.newton_maxit = (int32_t)(w >> 32);
    r.shape_type = bits & 15;
    r.ap_type = (bits >> 4) & 3;
    r.interaction = (bits >> 6) & 1;
    r.mat_type = (bits >> 7) & 1;
    r.aniso_class = (bits >> 8) & 3;
    r.frame_flags = (bits >> 10) & 7;
    r.curv = h0[1];
    r.cc = h0[2];
    r.g_shape[0] = h0[3]; r.g_shape[1] = h0[4]; r.g_shape[2] = h0[5];
    r.ap_p0 = h0[6];
    r.ap_p1 = h0[7];
    r.full = full;
}
PRT_DEV void unpack_hot_late(const prt_d8 &h1, hot_rec &r) {
    r.n_after = h1[0];
    r.aniso_eo = h1[1];
    r.aniso_ee = h1[2];
    r.aniso_axis[0] = h1[3]; r.aniso_axis[1] = h1[4]; r.aniso_axis[2] = h1[5];
}

// SHAPES: the shape code compiled in (as in k_trace_iso): tables whose surfaces are all conics get an
// instantiation without any Newton / polynomial / spline code.
// WANT_E: the caller stores the E fields behind the crystal interfaces (e_out); without it no eigenvector is computed
// for uniaxial / isotropic epsilon (prt_aniso.h).  uni: the uniform first segment (run-time flag: prologue only).
template <int MODE, bool GENERAL = true, bool PARK_LDS = false, bool WANT_E = false, int SHAPES = PRT_SHAPES_ALL>
__global__ __launch_bounds__(PRT_GENERAL_BLOCK, (!GENERAL && !WANT_E && SHAPES == PRT_SHAPES_CONIC) ? PRT_UNIAXIAL_WAVES : PRT_GENERAL_WAVES) void k_trace_general(
    const prt_dev_surface *__restrict__ tab, const prt_hot_surface *__restrict__ hot,
    const walk_step *__restrict__ walk, int32_t n_steps, int32_t A, int64_t N, int64_t in_pitch, int64_t P,
    const double *__restrict__ x0, const double *__restrict__ k0, const double *__restrict__ e_re,
    const double *__restrict__ e_im, int32_t e_mode, double *__restrict__ xh_out,
    double *__restrict__ k_out, double *__restrict__ e_out, double *__restrict__ e_out_im,
    uint8_t *__restrict__ valid_out_hit, uint8_t *__restrict__ valid_out_refr,
    uint8_t *__restrict__ nonconv_out = nullptr, first_uniform fu = first_uniform(), int32_t uni = 0) {
    const uint32_t tid = threadIdx.x;
    const int64_t blk = (int64_t)blockIdx.x * PRT_GENERAL_BLOCK;
    const int64_t i = blk + tid;
    if (i >= N) return;
    // N rays; P >= N is the ray pitch of the concatenated output layout (a bundle of P rays of which the last P - N
    // do not exist): with P a multiple of 16 every row of every level starts on a 128-B line -- 0.124 instead of
    // 0.151 ms on BASELINE configs[3], whose 998012 rays put every row at an odd multiple of 32 B
    vec3 x = v3(x0[i], x0[in_pitch + i], x0[2 * in_pitch + i]);
    vec3 k, d;
    if (uni) {  // uniform first segment: only x0 is read
        k = v3(fu.k[0], fu.k[1], fu.k[2]);
        d = uniform_first_direction(e_mode, fu, k);
    } else {
        k = v3(k0[i], k0[in_pitch + i], k0[2 * in_pitch + i]);
        vec3 kk[2] = {k, k};
        vec3 dd[2];
        first_direction<false>(e_mode, e_re, e_im, in_pitch, i, false, kk, dd);
        d = dd[0];
    }
    // the first walk entry and hot block (the following ones are fetched a step ahead, inside the loop)
    const PRT_CONST_AS volatile prt_i4 *wp = (const PRT_CONST_AS volatile prt_i4 *)(uint64_t)walk;
    prt_i4 w = wp[0];
    // all input loads land here, so that no wait inside the walk ever counts stores (PRT_WAIT_VMEM_LOADS)
    PRT_WAIT_VMEM_LOADS();
    bool valid = true;
    double d2 = 1.0;
    // per level: child 1 of the crystal interface of that level
    constexpr int PV = GENERAL ? 9 : 6;  // doubles per slot
    extern __shared__ double park_lds[];
    double parked[PARK_LDS ? 1 : PRT_FUSED_MAX_CRYSTALS][PV + 1];
    uint8_t *park_lds_alive = reinterpret_cast<uint8_t *>(park_lds + (size_t)A * PV * PRT_GENERAL_BLOCK);
    for (int32_t t = 0; t < n_steps; ++t) {
        // ---- this step's scalars; the next step's loads go out before anything is computed ----
        // ONE scalar-memory round trip per step: both halves of this step's hot block (the late half is not needed
        // before the interaction) and the NEXT walk entry, so that the next step knows its surface when it begins
        const int32_t s = w[0];
        const prt_d8 h0 = load_hot_half(hot, s, 0);
        const prt_d8 h1 = load_hot_half(hot, s, 1);
        const prt_i4 wn = wp[t + 1];  // (the program ends with a sentinel entry)
        const int32_t a = w[1] & 0xff, resume = (w[1] >> 8) & 0xff;
        const int64_t lp = (w[1] >> 16) & 0xff, cum_in = (MODE == PRT_MODE_PATH) ? w[2] : 0;
        const int64_t cum_out = (MODE == PRT_MODE_PATH) ? cum_in + (((int64_t)1 << a) - 1) : 0;
        const bool last_surface = (w[1] >> 24) != 0;
        hot_rec rec;
        unpack_hot_early(h0, tab + s, rec);
        const hot_rec *__restrict__ sf = &rec;
        if (resume) {  // take up the child parked at level resume - 1
            const int j = resume - 1;
            double pv[PV];
            uint8_t pb;
            if (PARK_LDS) {
                const double *slot = park_lds + (size_t)j * PV * PRT_GENERAL_BLOCK + threadIdx.x;
#pragma unroll
                for (int q = 0; q < PV; ++q) pv[q] = slot[q * PRT_GENERAL_BLOCK];
                pb = park_lds_alive[j * PRT_GENERAL_BLOCK + threadIdx.x];
            } else {
#pragma unroll
                for (int q = 0; q < PV; ++q) pv[q] = parked[j][q];
                pb = (uint8_t)parked[j][PV];
            }
            x = v3(pv[0], pv[1], pv[2]);
            valid = (pb & 1) != 0;
            if (GENERAL) {
                k = v3(pv[3], pv[4], pv[5]);
                d = v3(pv[PV - 3], pv[PV - 2], pv[PV - 1]);
            } else {  // rebuild k and d from the parked wave vector, with the record of the parking surface
                const int32_t sp = w[3];
                hot_rec prec;
                unpack_hot_early(load_hot_half(hot, sp, 0), tab + sp, prec);
                unpack_hot_late(load_hot_half(hot, sp, 1), prec);
                closed_form_ray(&prec, v3(pv[3], pv[4], pv[5]), (pb & 2) != 0, k, d);
            }
            d2 = 1.0;
        }
        const bool store = (MODE == PRT_MODE_PATH || last_surface);
        const bool crystal = sf->mat_type == PRT_MAT_ANISOTROPIC;
        const int64_t n_in = P << a;
        const int a_out = crystal ? a + 1 : a;
        const int64_t n_out = P << a_out;
        // store addresses = wave-uniform row base (scalar registers) + the thread's 32-bit offset: the
        // form global_store takes with a scalar base, no 64-bit vector arithmetic per store
        const int64_t row = blk + P * lp;
        const bool alive = valid;
        vec3 xh, p, g;
        double g2;
        bool ncv;
        propagate_step<SHAPES>(sf, x, d, d2, xh, p, g, g2, valid, ncv);
        if (store) {
            PRT_GLOBAL_AS double *xrow = uniform_ptr(xh_out + 3 * P * cum_in + row);
            PRT_GSTORE(xrow + tid, xh.x);
            PRT_GSTORE(xrow + n_in + tid, xh.y);
            PRT_GSTORE(xrow + 2 * n_in + tid, xh.z);
            PRT_GSTORE_MASK(uniform_ptr(valid_out_hit + P * cum_in + row) + tid, (uint8_t)(valid ? 1 : 0));
            if (nonconv_out) uniform_ptr(nonconv_out + P * cum_in + row)[tid] = ncv ? 1 : 0;
        }
        x = xh;
        unpack_hot_late(h1, rec);
        if (crystal) {
            aniso_solution sol[2];
            // The normal: from the gradient the intersection left behind (for a sphere it IS the unit normal) -- unless
            // the E fields are wanted: for eps = e I they are an arbitrary basis picked by comparisons of k's components,
            // and the per-surface entry point, which evaluates the normal from the hit point, must pick the same one.
            if (WANT_E) interact_anisotropic<GENERAL, SHAPES>(sf, p, k, sol, true);
            else interact_anisotropic_n<GENERAL>(sf, normal_from_grad<SHAPES>(sf, g, g2), k, sol, false);
            valid = alive;  // no validity filtering at a crystal interface (ray.py:68)
            if (store) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int64_t row_out = row + ((P * b) << a);
                    PRT_GLOBAL_AS double *krow = uniform_ptr(k_out + 3 * P * cum_out + row_out);
                    PRT_GSTORE(krow + tid, sol[b].k.x);
                    PRT_GSTORE(krow + n_out + tid, sol[b].k.y);
                    PRT_GSTORE(krow + 2 * n_out + tid, sol[b].k.z);
                    if (WANT_E && e_out) {
                        PRT_GLOBAL_AS double *eo = uniform_ptr(e_out + 3 * P * cum_out + row_out);
                        eo[tid] = sol[b].er.x;
                        (eo + n_out)[tid] = sol[b].er.y;
                        (eo + 2 * n_out)[tid] = sol[b].er.z;
                        if (e_out_im) {
                            PRT_GLOBAL_AS double *ei = uniform_ptr(e_out_im + 3 * P * cum_out + row_out);
                            ei[tid] = sol[b].ei.x;
                            (ei + n_out)[tid] = sol[b].ei.y;
                            (ei + 2 * n_out)[tid] = sol[b].ei.z;
                        }
                    }
                    if (valid_out_refr) PRT_GSTORE_MASK(uniform_ptr(valid_out_refr + P * cum_out + row_out) + tid, (uint8_t)(alive ? 1 : 0));
                }
            }
            {
                double pv[PV];
                pv[0] = xh.x; pv[1] = xh.y; pv[2] = xh.z;
                uint8_t pb = alive ? 1 : 0;
                if (GENERAL) {
                    pv[3] = sol[1].k.x; pv[4] = sol[1].k.y; pv[5] = sol[1].k.z;
                    pv[PV - 3] = sol[1].d.x; pv[PV - 2] = sol[1].d.y; pv[PV - 1] = sol[1].d.z;
                } else {
                    pv[3] = sol[1].kv.x; pv[4] = sol[1].kv.y; pv[5] = sol[1].kv.z;
                    pb |= sol[1].is_e ? 2 : 0;
                }
                if (PARK_LDS) {
                    double *slot = park_lds + (size_t)a * PV * PRT_GENERAL_BLOCK + threadIdx.x;
#pragma unroll
                    for (int q = 0; q < PV; ++q) slot[q * PRT_GENERAL_BLOCK] = pv[q];
                    park_lds_alive[a * PRT_GENERAL_BLOCK + threadIdx.x] = pb;
                } else {
#pragma unroll
                    for (int q = 0; q < PV; ++q) parked[a][q] = pv[q];
                    parked[a][PV] = (double)pb;
                }
            }
            k = sol[0].k;
            d = sol[0].d;
            d2 = 1.0;
        } else {
            const vec3 n = normal_from_grad<SHAPES>(sf, g, g2);
            interact_isotropic(sf, n, k, valid);
            d = k;
            d2 = sf->n_after * sf->n_after;
            if (store) {
                PRT_GLOBAL_AS double *krow = uniform_ptr(k_out + 3 * P * cum_out + row);
                PRT_GSTORE(krow + tid, k.x);
                PRT_GSTORE(krow + n_out + tid, k.y);
                PRT_GSTORE(krow + 2 * n_out + tid, k.z);
                if (valid_out_refr) PRT_GSTORE_MASK(uniform_ptr(valid_out_refr + P * cum_out + row) + tid, (uint8_t)(valid ? 1 : 0));
            }
        }
        w = wn;
    }
}

// ---------------------------------------------------------------------------
// Evanescent modes at a crystal interface: the reference carries them as COMPLEX wave vectors (complex xi from
// LAPACK, material.py:407-454; k = kpa + xi n, material_anisotropic.py:87-100); the march computes in real
// arithmetic and leaves NaN in such a slot (the mode carries no energy through the interface and is not traced
// further).  This post-pass over one crystal surface of a finished PATH-mode trace fills those slots: real part
// into k_re, imaginary part into k_im (0 everywhere else).  Which root lands in which slot follows the order the
// march (and the reference's S.n sort, where an evanescent mode has S.n = 0) gives the two leaving solutions:
// uniaxial / isotropic eps -- closed forms, ordinary before extraordinary; general eps -- the complex roots of the
// quartic (Aberth) with positive imaginary part, by ascending real part.  Of a conjugate pair xi, conj(xi) the one
// with Im > 0 is reported (the reference's pick is whatever its sort leaves first).
//   x_hit_s (3, BP) hit points of the surface; k_par: wave vectors of the entering rays, rows par_pitch apart,
//   indexed by slot (NULL: every ray has fu.k); k_re / k_im (3, 2 BP): the surface's block of k_out / k_out_im;
//   BP = branches * P slots, ray i of a branch exists for i < N.
__global__ __launch_bounds__(PRT_BLOCK) void k_evanescent_fill(
    const prt_dev_surface *__restrict__ sf, int64_t N, int64_t P, int64_t BP, const double *__restrict__ x_hit_s,
    const double *__restrict__ k_par, int64_t par_pitch, first_uniform fu, double *__restrict__ k_re,
    double *__restrict__ k_im) {
    const int64_t j = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    bool live = j < BP && (j % P) < N;
    const int64_t jj = live ? j : 0;
    const int64_t M = 2 * BP;
    const bool nan0 = live && isnan(k_re[jj]), nan1 = live && isnan(k_re[jj + BP]);
    live = live && (nan0 || nan1);
    // A wave without an evanescent mode leaves together, before it has read anything but the two wave-vector
    // components above: on a path without total reflection the pass costs 16 B per entering ray instead of 64
    // (configs[3] through the drop-in layer: 48 -> ~10 us per crystal surface).  (No per-lane return: quartic_roots
    // votes across the wave.)
    if (!__any(live)) return;
    const vec3 xh = v3(x_hit_s[jj], x_hit_s[BP + jj], x_hit_s[2 * BP + jj]);
    const vec3 kg = k_par ? v3(k_par[jj], k_par[par_pitch + jj], k_par[2 * par_pitch + jj]) : v3(fu.k[0], fu.k[1], fu.k[2]);
    const vec3 n = normal_in_material_frame(sf, to_shape_frame(sf, xh));
    const bool mat_id = sf->frame_flags & PRT_FRAME_MAT_IDENTITY;
    const vec3 k1 = mat_id ? kg : matT_vec(sf->B_mat, kg);
    const double kn = dot(k1, n);
    const vec3 kpa = v3(k1.x - kn * n.x, k1.y - kn * n.y, k1.z - kn * n.z);
    const double kap2 = dot(kpa, kpa);
    live = live && isfinite(kap2) && isfinite(n.x) && isfinite(n.y) && isfinite(n.z);
    cplx cand[2] = {cplx{0.0, 0.0}, cplx{0.0, 0.0}};
    int nc = 0;
    const int cls = sf->aniso_class;
    if (cls == PRT_ANISO_ISOTROPIC) {
        const double q = sf->aniso_eo - kap2;
        if (q < 0.0) {
            cand[0] = cand[1] = cplx{0.0, sqrt(-q)};
            nc = 2;
        }
    } else if (cls == PRT_ANISO_UNIAXIAL) {
        const double eo = sf->aniso_eo, ee = sf->aniso_ee;
        const vec3 c = v3(sf->aniso_axis[0], sf->aniso_axis[1], sf->aniso_axis[2]);
        const double q = eo - kap2;
        if (q < 0.0) cand[nc++] = cplx{0.0, sqrt(-q)};
        const double ncx = dot(n, c), kc = dot(kpa, c);
        const double A = eo + (ee - eo) * ncx * ncx;
        const double Bh = (ee - eo) * kc * ncx;
        const double C = eo * kap2 + (ee - eo) * kc * kc - eo * ee;
        const double disc = Bh * Bh - A * C;
        if (disc < 0.0) cand[nc++] = cplx{-Bh / A, sqrt(-disc) / fabs(A)};
    } else {
        double pc[5];
        cplx z[4];
        xi_polynomial(sf->eps_re, n, kpa, pc);
        if (!live) {  // keep the iteration of dead lanes harmless
            pc[0] = -1.0; pc[1] = 0.0; pc[2] = 0.0; pc[3] = 0.0; pc[4] = 1.0;
        }
        quartic_roots(pc, z);
        // the complex roots with positive imaginary part, by ascending real part
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool cx = z[i].im > 1e-9 * fmax(1.0, fabs(z[i].re));
            if (cx && nc < 2) cand[nc++] = z[i];
        }
        if (nc == 2 && cand[1].re < cand[0].re) {
            const cplx t = cand[0];
            cand[0] = cand[1];
            cand[1] = t;
        }
    }
    if (!live) return;
    const double sg = (sf->interaction == PRT_MIRROR) ? -1.0 : 1.0;  // reflect: -(k), material_anisotropic.py:136
    int used = 0;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (!(b == 0 ? nan0 : nan1) || used >= nc) continue;
        const cplx xi = cand[used++];
        vec3 kr = v3(sg * (kpa.x + xi.re * n.x), sg * (kpa.y + xi.re * n.y), sg * (kpa.z + xi.re * n.z));
        vec3 ki = v3(sg * xi.im * n.x, sg * xi.im * n.y, sg * xi.im * n.z);
        if (!mat_id) {
            kr = mat_vec(sf->B_mat, kr);
            ki = mat_vec(sf->B_mat, ki);
        }
        const int64_t o = jj + b * BP;
        k_re[o] = kr.x; k_re[M + o] = kr.y; k_re[2 * M + o] = kr.z;
        k_im[o] = ki.x; k_im[M + o] = ki.y; k_im[2 * M + o] = ki.z;
    }
}

// ---------------------------------------------------------------------------
// per-surface kernels (the plugin-granular API, and the march through
// anisotropic systems).  x is read modulo n_src so that the two children of a
// split ray share their parent's hit point without a copy.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(PRT_BLOCK) void k_propagate(
    const prt_dev_surface *__restrict__ sf, int64_t N, int64_t n_src,
    const double *__restrict__ x_in, const double *__restrict__ k_in,
    const double *__restrict__ dir_in, const double *__restrict__ e_re,
    const double *__restrict__ e_im, int32_t e_mode, const uint8_t *__restrict__ valid_in,
    double *__restrict__ xh_out, uint8_t *__restrict__ valid_out, uint8_t *__restrict__ nonconv_out = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const int64_t j = (n_src == N) ? i : (i % n_src);
    const vec3 x = v3(x_in[j], x_in[n_src + j], x_in[2 * n_src + j]);
    vec3 d;
    if (dir_in) {
        d = v3(dir_in[i], dir_in[N + i], dir_in[2 * N + i]);
    } else {
        const vec3 kk = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
        vec3 k[2] = {kk, kk};
        vec3 dd[2];
        first_direction<false>(e_mode, e_re, e_im, N, i, false, k, dd);
        d = dd[0];
    }
    bool valid = valid_in ? (valid_in[i] != 0) : true;
    vec3 xh, p, g;
    double g2;
    bool ncv;
    propagate_step(sf, x, d, 1.0, xh, p, g, g2, valid, ncv);
    xh_out[i] = xh.x;
    xh_out[N + i] = xh.y;
    xh_out[2 * N + i] = xh.z;
    valid_out[i] = valid ? 1 : 0;
    if (nonconv_out) nonconv_out[i] = ncv ? 1 : 0;
}

__global__ __launch_bounds__(PRT_BLOCK) void k_interact_iso(
    const prt_dev_surface *__restrict__ sf, int64_t N, const double *__restrict__ xh_in,
    const double *__restrict__ k_in, const uint8_t *__restrict__ valid_in,
    double *__restrict__ k_out, double *__restrict__ dir_out, uint8_t *__restrict__ valid_out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const vec3 xh = v3(xh_in[i], xh_in[N + i], xh_in[2 * N + i]);
    vec3 k = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
    bool valid = valid_in ? (valid_in[i] != 0) : true;
    const vec3 p = to_shape_frame(sf, xh);
    interact_isotropic(sf, normal_in_material_frame(sf, p), k, valid);
    k_out[i] = k.x;
    k_out[N + i] = k.y;
    k_out[2 * N + i] = k.z;
    if (dir_out) {
        const vec3 d = normalized(k);
        dir_out[i] = d.x;
        dir_out[N + i] = d.y;
        dir_out[2 * N + i] = d.z;
    }
    if (valid_out) valid_out[i] = valid ? 1 : 0;
}

// ---------------------------------------------------------------------------
// The same two steps for BIG isotropic bundles (prt_propagate_rows / prt_interact_rows): row-pitched arrays, a thread
// owns two adjacent rays (16 B per lane and access: one dwordx4 per row like the fused march), non-temporal hints --
// every array is read once and written once.  Bytes per ray and surface: propagate 49 in (x, k, mask) + 25 out
// (x_hit, mask), interact 49 in + 25 out (k_out, mask): 148 B, against the 195 B of the tight one-ray-per-thread
// kernels above when they also write the ray direction.  After an isotropic interaction the ray direction is parallel
// to k (E is perpendicular to k: ray.py:136-152), and conic_t / explicit_t are homogeneous in d, so the step takes the
// unnormalised k with d2 = k.k -- what the fused march does (k_trace_iso).
// (Tried in round 6 and removed: each kernel walking the bundle in the opposite direction of its predecessor, so that it
// starts with the lines touched last -- memory-side cache, 256 MB against 740 MB per kernel: 2.954 -> 2.947 ms per sweep,
// nothing: profiles/r06d_ab_plugin_traversal_and_placement.txt.)
// ---------------------------------------------------------------------------
template <bool VEC>
PRT_DEV void load_mask2(const uint8_t *__restrict__ m, int64_t i, bool second, bool b[2]) {
    if (!m) {
        b[0] = b[1] = true;
    } else if (VEC && second) {
        const unsigned v = *reinterpret_cast<const uint16_t *>(m + i);
        b[0] = (v & 0xFFu) != 0;
        b[1] = (v >> 8) != 0;
    } else {
        b[0] = m[i] != 0;
        b[1] = second ? (m[i + 1] != 0) : b[0];
    }
}

template <bool VEC>
PRT_DEV void store_mask2(uint8_t *__restrict__ m, int64_t i, bool second, const bool b[2]) {
    if (VEC && second) {         // (the byte behind an odd bundle's last ray is not ours)
        *reinterpret_cast<uint16_t *>(m + i) = (uint16_t)((b[0] ? 1u : 0u) | (b[1] ? 0x100u : 0u));
    } else {
        m[i] = b[0] ? 1 : 0;
        if (second) m[i + 1] = b[1] ? 1 : 0;
    }
}

// SHAPES: the shape code compiled in (the surface's own: conics alone need half the registers of the general case)
template <bool VEC, int SHAPES>
__global__ __launch_bounds__(PRT_MARCH_BLOCK) void k_propagate_rows(
    const prt_dev_surface *__restrict__ sf, int64_t N, const double *__restrict__ x_in, int64_t x_pitch,
    const double *__restrict__ k_in, int64_t k_pitch, const double *__restrict__ dir_in,
    const double *__restrict__ e_re, const double *__restrict__ e_im, int32_t e_mode,
    const uint8_t *__restrict__ valid_in, double *__restrict__ xh_out, int64_t out_pitch,
    uint8_t *__restrict__ valid_out, uint8_t *__restrict__ nonconv_out) {
    const int64_t i = ((int64_t)blockIdx.x * PRT_MARCH_BLOCK + threadIdx.x) * 2;
    if (i >= N) return;
    const bool second = (i + 1 < N);
    vec3 x[2], d[2];
    double d2[2] = {1.0, 1.0};
    rayio<VEC>::load(x_in, x_pitch, i, second, x);
    if (dir_in) {
        rayio<VEC>::load(dir_in, k_pitch, i, second, d);
    } else {
        vec3 k[2];
        rayio<VEC>::load(k_in, k_pitch, i, second, k);
        if (e_mode == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                d[r] = k[r];
                d2[r] = dot(k[r], k[r]);
            }
        } else {
            first_direction<VEC>(e_mode, e_re, e_im, k_pitch, i, second, k, d);
        }
    }
    bool valid[2], ncv[2];
    load_mask2<VEC>(valid_in, i, second, valid);
    PRT_WAIT_VMEM_LOADS();
    vec3 xh[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        vec3 p, g;
        double g2;
        propagate_step<SHAPES>(sf, x[r], d[r], d2[r], xh[r], p, g, g2, valid[r], ncv[r]);
    }
    rayio<VEC>::store(xh_out, out_pitch, i, second, xh);
    store_mask2<VEC>(valid_out, i, second, valid);
    if (nonconv_out) store_mask2<VEC>(nonconv_out, i, second, ncv);
}

template <bool VEC, int SHAPES>
__global__ __launch_bounds__(PRT_MARCH_BLOCK) void k_interact_iso_rows(
    const prt_dev_surface *__restrict__ sf, int64_t N, const double *__restrict__ xh_in, int64_t x_pitch,
    const double *__restrict__ k_in, int64_t k_pitch, const uint8_t *__restrict__ valid_in,
    double *__restrict__ k_out, int64_t out_pitch, double *__restrict__ dir_out,
    uint8_t *__restrict__ valid_out) {
    const int64_t i = ((int64_t)blockIdx.x * PRT_MARCH_BLOCK + threadIdx.x) * 2;
    if (i >= N) return;
    const bool second = (i + 1 < N);
    vec3 xh[2], k[2];
    bool valid[2];
    rayio<VEC>::load(xh_in, x_pitch, i, second, xh);
    rayio<VEC>::load(k_in, k_pitch, i, second, k);
    load_mask2<VEC>(valid_in, i, second, valid);
    PRT_WAIT_VMEM_LOADS();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const vec3 p = to_shape_frame(sf, xh[r]);
        interact_isotropic(sf, normal_in_material_frame<SHAPES>(sf, p), k[r], valid[r]);
    }
    rayio<VEC>::store(k_out, out_pitch, i, second, k);
    if (dir_out) {
        vec3 d[2] = {normalized(k[0]), normalized(k[1])};
        rayio<VEC>::store(dir_out, out_pitch, i, second, d);
    }
    if (valid_out) store_mask2<VEC>(valid_out, i, second, valid);
}

// ONE SURFACE of the march as a launch of its own (prt_surface_step_rows): Material.propagate + Material.refract | reflect
// at a surface with an isotropic, lossless medium behind it -- the loop body of OpticalElement.seqtrace
// (optical_element.py:336-375) and SURVEY section 2's K1 / K2 kernels in their per-surface form.  Reads the 49-B state
// (x, k, mask), writes the 49-B record (x_hit, k_out, masks): SURVEY 8d's 98 B per ray-surface-op.  The arithmetic is
// the fused march's (k_trace_iso: the normal comes from the gradient the intersection left behind).
template <bool VEC, int SHAPES>
__global__ __launch_bounds__(PRT_MARCH_BLOCK) void k_surface_step_rows(
    const prt_dev_surface *__restrict__ sf, int64_t N, const double *__restrict__ x_in, int64_t x_pitch,
    const double *__restrict__ k_in, int64_t k_pitch, const double *__restrict__ dir_in,
    const double *__restrict__ e_re, const double *__restrict__ e_im, int32_t e_mode,
    const uint8_t *__restrict__ valid_in, double *__restrict__ xh_out, double *__restrict__ k_out, int64_t out_pitch,
    uint8_t *__restrict__ valid_hit, uint8_t *__restrict__ valid_out, uint8_t *__restrict__ nonconv_out) {
    const int64_t i = ((int64_t)blockIdx.x * PRT_MARCH_BLOCK + threadIdx.x) * 2;
    if (i >= N) return;
    const bool second = (i + 1 < N);
    vec3 x[2], k[2], d[2];
    double d2[2] = {1.0, 1.0};
    rayio<VEC>::load(x_in, x_pitch, i, second, x);
    rayio<VEC>::load(k_in, k_pitch, i, second, k);
    if (dir_in) {
        rayio<VEC>::load(dir_in, k_pitch, i, second, d);
    } else if (e_mode == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            d[r] = k[r];
            d2[r] = dot(k[r], k[r]);
        }
    } else {
        first_direction<VEC>(e_mode, e_re, e_im, k_pitch, i, second, k, d);
    }
    bool valid[2], vhit[2], ncv[2];
    load_mask2<VEC>(valid_in, i, second, valid);
    PRT_WAIT_VMEM_LOADS();
    vec3 xh[2], nrm[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        vec3 p, g;
        double g2;
        propagate_step<SHAPES>(sf, x[r], d[r], d2[r], xh[r], p, g, g2, valid[r], ncv[r]);
        vhit[r] = valid[r];
        nrm[r] = normal_from_grad<SHAPES>(sf, g, g2);
    }
    rayio<VEC>::store(xh_out, out_pitch, i, second, xh);          // (the hit points go out before the interaction: k_trace_iso)
#pragma unroll
    for (int r = 0; r < 2; ++r) interact_isotropic(sf, nrm[r], k[r], valid[r]);
    rayio<VEC>::store(k_out, out_pitch, i, second, k);
    store_mask2<VEC>(valid_hit, i, second, vhit);
    store_mask2<VEC>(valid_out, i, second, valid);
    if (nonconv_out) store_mask2<VEC>(nonconv_out, i, second, ncv);
}

__global__ __launch_bounds__(PRT_BLOCK) void k_interact_aniso(
    const prt_dev_surface *__restrict__ sf, int64_t N, const double *__restrict__ xh_in,
    const double *__restrict__ k_in, const uint8_t *__restrict__ alive_in,
    double *__restrict__ k_out, double *__restrict__ dir_out, double *__restrict__ e_re_out,
    double *__restrict__ e_im_out, uint8_t *__restrict__ valid_out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const vec3 xh = v3(xh_in[i], xh_in[N + i], xh_in[2 * N + i]);
    const vec3 k = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
    const vec3 p = to_shape_frame(sf, xh);
    // The reference's anisotropic refract does no validity filtering: every ray that is
    // still IN the bundle gets two children in a fresh all-valid bundle (ray.py:68).
    // In the dense representation "in the bundle" = alive_in (the mask the previous
    // compaction used); rays compacted away earlier must stay dead.
    const uint8_t alive = alive_in ? alive_in[i] : (uint8_t)1;
    aniso_solution sol[2];
    interact_anisotropic(sf, p, k, sol, e_re_out != nullptr);
    const int64_t M = 2 * N;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int64_t o = i + b * N;  // np.hstack((sol2, sol3)), material_anisotropic.py:89
        k_out[o] = sol[b].k.x;
        k_out[M + o] = sol[b].k.y;
        k_out[2 * M + o] = sol[b].k.z;
        dir_out[o] = sol[b].d.x;
        dir_out[M + o] = sol[b].d.y;
        dir_out[2 * M + o] = sol[b].d.z;
        if (e_re_out) {
            e_re_out[o] = sol[b].er.x;
            e_re_out[M + o] = sol[b].er.y;
            e_re_out[2 * M + o] = sol[b].er.z;
        }
        if (e_im_out) {
            e_im_out[o] = sol[b].ei.x;
            e_im_out[M + o] = sol[b].ei.y;
            e_im_out[2 * M + o] = sol[b].ei.z;
        }
        if (valid_out) valid_out[o] = alive;
    }
}

// The same interface for tables with a complex (absorbing) epsilon tensor: complex wave vectors in and out
// (k_im_in may be NULL: the ray comes from a lossless medium), prt_aniso_cplx.h.  eps_im: the imaginary part of
// this surface's tensor (9 doubles, device memory; not part of the march's record).
__global__ __launch_bounds__(PRT_BLOCK) void k_interact_aniso_cplx(
    const prt_dev_surface *__restrict__ sf, const double *__restrict__ eps_im, int64_t N,
    const double *__restrict__ xh_in, const double *__restrict__ k_in, const double *__restrict__ k_im_in,
    const uint8_t *__restrict__ alive_in, double *__restrict__ k_out, double *__restrict__ k_im_out,
    double *__restrict__ dir_out, double *__restrict__ e_re_out, double *__restrict__ e_im_out,
    uint8_t *__restrict__ valid_out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const vec3 xh = v3(xh_in[i], xh_in[N + i], xh_in[2 * N + i]);
    const vec3 kr = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
    const vec3 ki = k_im_in ? v3(k_im_in[i], k_im_in[N + i], k_im_in[2 * N + i]) : v3(0.0, 0.0, 0.0);
    const vec3 p = to_shape_frame(sf, xh);
    const uint8_t alive = alive_in ? alive_in[i] : (uint8_t)1;
    aniso_solution_cplx sol[2];
    interact_anisotropic_cplx(sf, eps_im, normal_in_material_frame(sf, p), kr, ki, sol);
    const int64_t M = 2 * N;
    for (int b = 0; b < 2; ++b) {
        const int64_t o = i + b * N;
        k_out[o] = sol[b].k_re.x;
        k_out[M + o] = sol[b].k_re.y;
        k_out[2 * M + o] = sol[b].k_re.z;
        k_im_out[o] = sol[b].k_im.x;
        k_im_out[M + o] = sol[b].k_im.y;
        k_im_out[2 * M + o] = sol[b].k_im.z;
        dir_out[o] = sol[b].d.x;
        dir_out[M + o] = sol[b].d.y;
        dir_out[2 * M + o] = sol[b].d.z;
        if (e_re_out) {
            e_re_out[o] = sol[b].e_re.x;
            e_re_out[M + o] = sol[b].e_re.y;
            e_re_out[2 * M + o] = sol[b].e_re.z;
        }
        if (e_im_out) {
            e_im_out[o] = sol[b].e_im.x;
            e_im_out[M + o] = sol[b].e_im.y;
            e_im_out[2 * M + o] = sol[b].e_im.z;
        }
        if (valid_out) valid_out[o] = alive;
    }
}

// IsotropicMaterial.refract / reflect (material_isotropic.py:137-236) with a complex incoming wave vector and / or
// a complex refractive index n_re + i n_im behind the surface -- the LAST surface of a table with absorbing media:
//   kin = k1 - (k1.n) n,  square = n^2 - kin.kin,  valid = square > 0 (NumPy's order of complex numbers: by real part,
//   then imaginary part),  xi = sqrt(square) (principal branch),  k2 = +-kin + xi n.
__global__ __launch_bounds__(PRT_BLOCK) void k_interact_iso_cplx(
    const prt_dev_surface *__restrict__ sf, double n_im, int64_t N, const double *__restrict__ xh_in,
    const double *__restrict__ k_in, const double *__restrict__ k_im_in, const uint8_t *__restrict__ valid_in,
    double *__restrict__ k_out, double *__restrict__ k_im_out, uint8_t *__restrict__ valid_out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const vec3 xh = v3(xh_in[i], xh_in[N + i], xh_in[2 * N + i]);
    const vec3 kr = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
    const vec3 ki = k_im_in ? v3(k_im_in[i], k_im_in[N + i], k_im_in[2 * N + i]) : v3(0.0, 0.0, 0.0);
    bool valid = valid_in ? (valid_in[i] != 0) : true;
    const vec3 n = normal_in_material_frame(sf, to_shape_frame(sf, xh));
    const cvec3 k1 = cv3(matT_vec(sf->B_mat, kr), matT_vec(sf->B_mat, ki));
    const cx kn = cv_dot(k1, n);
    const cvec3 kin = cvec3{k1.x - n.x * kn, k1.y - n.y * kn, k1.z - n.z * kn};
    const cx nn = cx{sf->n_after, n_im};
    const cx sq = nn * nn - cv_dot(kin, kin);
    const bool ok = sq.re > 0.0 || (sq.re == 0.0 && sq.im > 0.0);
    // principal square root
    const double r = sqrt(sq.re * sq.re + sq.im * sq.im);
    cx xi;
    if (sq.re >= 0.0) {
        xi.re = sqrt(0.5 * (r + sq.re));
        xi.im = xi.re > 0.0 ? sq.im / (2.0 * xi.re) : 0.0;
    } else {
        xi.im = copysign(sqrt(0.5 * (r - sq.re)), sq.im);
        xi.re = sq.im / (2.0 * xi.im);
    }
    valid = valid && ok && isfinite(n.x) && isfinite(n.y) && isfinite(n.z);
    const double sgn = sf->interaction == PRT_MIRROR ? -1.0 : 1.0;
    const cvec3 k2 = cvec3{sgn * kin.x + n.x * xi, sgn * kin.y + n.y * xi, sgn * kin.z + n.z * xi};
    const vec3 o_re = mat_vec(sf->B_mat, cv_re(k2)), o_im = mat_vec(sf->B_mat, cv_im(k2));
    k_out[i] = o_re.x;
    k_out[N + i] = o_re.y;
    k_out[2 * N + i] = o_re.z;
    if (k_im_out) {
        k_im_out[i] = o_im.x;
        k_im_out[N + i] = o_im.y;
        k_im_out[2 * N + i] = o_im.z;
    }
    if (valid_out) valid_out[i] = valid ? 1 : 0;
}

__global__ __launch_bounds__(PRT_BLOCK) void k_shape_eval(const prt_dev_surface *__restrict__ sf,
                                                          int64_t N, const double *__restrict__ x,
                                                          const double *__restrict__ y,
                                                          double *__restrict__ sag,
                                                          double *__restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const double xx = x[i], yy = y[i];
    if (sag) sag[i] = shape_sag(sf, xx, yy);
    if (grad) {
        vec3 g;
        if (sf->shape_type == PRT_SHAPE_CONIC) {
            // Conic.getGrad as the reference evaluates it (surface_shape.py:229-235)
            const double z = conic_sag(sf->curv, sf->cc, xx * xx + yy * yy);
            g = v3(-sf->curv * xx, -sf->curv * yy, 1.0 - sf->curv * z * (1.0 + sf->cc));
        } else {
            g = shape_grad(sf, xx, yy);
        }
        grad[i] = g.x;
        grad[N + i] = g.y;
        grad[2 * N + i] = g.z;
    }
}

// A unit E field perpendicular to k for bundles leaving an isotropic interface.  The
// reference takes the singular vector of the smallest singular value of
// -k^2 I + k k^T + n^2 I (material_isotropic.py:72-128), which is an ARBITRARY unit
// vector of the 2-d null space {E : E.k = 0}; this picks E = unit(k x a), a = the
// coordinate axis least aligned with k.  Not on the x/k parity contract.
__global__ __launch_bounds__(PRT_BLOCK) void k_efield_perp(int64_t N, const double *__restrict__ k_in,
                                                           double *__restrict__ e_out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const vec3 k = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
    const double ax = fabs(k.x), ay = fabs(k.y), az = fabs(k.z);
    const vec3 a = (ay <= ax && ay <= az) ? v3(0, 1, 0) : ((ax <= az) ? v3(1, 0, 0) : v3(0, 0, 1));
    const vec3 e = normalized(cross(k, a));
    e_out[i] = e.x;
    e_out[N + i] = e.y;
    e_out[2 * N + i] = e.z;
}

// RayBundle.returnKtoD (ray.py:136-152) for one stored point: unit Poynting direction from
// (k, E); e_mode as in first_direction (0: k/|k|, 1: E = ey, 2: E given).
__global__ __launch_bounds__(PRT_BLOCK) void k_poynting_dir(int64_t N, const double *__restrict__ k_in,
                                                            const double *__restrict__ e_re,
                                                            const double *__restrict__ e_im,
                                                            int32_t e_mode,
                                                            double *__restrict__ d_out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const vec3 kk = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
    vec3 k[2] = {kk, kk};
    vec3 dd[2];
    first_direction<false>(e_mode, e_re, e_im, N, i, false, k, dd);
    d_out[i] = dd[0].x;
    d_out[N + i] = dd[0].y;
    d_out[2 * N + i] = dd[0].z;
}

// RayBundleAnalysis.get_arc_length / get_phase_difference (analysis/ray_analysis.py:136-163):
// per ray, sum over consecutive stored points p of |x_{p+1} - x_p|  (mode 0) or of
// x_{p+1}.k_{p+1} - x_p.k_p (mode 1).  xs / ks: device tables of P pointers to tight (3,N) arrays.
__global__ __launch_bounds__(PRT_BLOCK) void k_path_sums(int32_t P, int64_t N,
                                                         const double *const *__restrict__ xs,
                                                         const double *const *__restrict__ ks,
                                                         int32_t mode, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    double acc = 0.0;
    vec3 xa = v3(xs[0][i], xs[0][N + i], xs[0][2 * N + i]);
    double pa = 0.0;
    if (mode == 1) pa = dot(xa, v3(ks[0][i], ks[0][N + i], ks[0][2 * N + i]));
    for (int p = 1; p < P; ++p) {
        const vec3 xb = v3(xs[p][i], xs[p][N + i], xs[p][2 * N + i]);
        if (mode == 0) {
            const vec3 dlt = v3(xb.x - xa.x, xb.y - xa.y, xb.z - xa.z);
            acc += sqrt(dot(dlt, dlt));
        } else {
            const double pb = dot(xb, v3(ks[p][i], ks[p][N + i], ks[p][2 * N + i]));
            acc += pb - pa;
            pa = pb;
        }
        xa = xb;
    }
    out[i] = acc;
}

// ---------------------------------------------------------------------------
// bundle moments: count, sum (x - ref), sum (x - ref)^2 per component over the rays whose
// mask byte is non-zero (all rays if mask == NULL).  Two deterministic stages (fixed
// summation order, no atomics): grid-stride partials per block, then one block adds the
// partials.  Feeds RayBundleAnalysis.get_centroid_position / get_rms_spot_size
// (analysis/ray_analysis.py:44-86) and turns the multi-GPU image-plane exchange into a
// 7-double all-reduce.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(PRT_BLOCK) void k_moments_partial(int64_t N, int64_t pitch,
                                                               const double *__restrict__ x,
                                                               const uint8_t *__restrict__ mask,
                                                               int32_t mode, double rx, double ry,
                                                               double rz,
                                                               const double *__restrict__ ref_dev,
                                                               int32_t ref_kind,
                                                               double *__restrict__ partials,
                                                               int32_t mask_bits = 0xff) {
    // mask_bits: which bits of a mask byte select a ray (0xff: any; 2: the valid_out bit of packed flags)
    // ref_kind 1: ref_dev holds the reference point (3 doubles); 2: ref_dev holds a moments
    // vector {count, sum x, ...} (e.g. all-reduced over the ranks) -> reference = centroid
    if (ref_kind == 1) {
        rx = ref_dev[0];
        ry = ref_dev[1];
        rz = ref_dev[2];
    } else if (ref_kind == 2) {
        const double inv = 1.0 / (ref_dev[0] + 1e-17);  // numerical_tolerance, ray_analysis.py:55
        rx = ref_dev[1] * inv;
        ry = ref_dev[2] * inv;
        rz = ref_dev[3] * inv;
    }
    double acc[MOM_VALUES] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x; i < N;
         i += (int64_t)gridDim.x * PRT_BLOCK) {
        if (mask && !(mask[i] & mask_bits)) continue;
        double dx = x[i], dy = x[pitch + i], dz = x[2 * pitch + i];
        if (mode == 0) {  // points relative to ref
            dx -= rx;
            dy -= ry;
            dz -= rz;
        } else {  // unit directions (ray.py:136-152 for E perpendicular to k), optionally x ref
            const vec3 u = normalized(v3(dx, dy, dz));
            dx = u.x;
            dy = u.y;
            dz = u.z;
            if (mode == 2) {
                const vec3 c = cross(u, v3(rx, ry, rz));
                dx = c.x;
                dy = c.y;
                dz = c.z;
            }
        }
        acc[0] += 1.0;
        acc[1] += dx;
        acc[2] += dy;
        acc[3] += dz;
        acc[4] += dx * dx;
        acc[5] += dy * dy;
        acc[6] += dz * dz;
    }
    __shared__ double sh[PRT_BLOCK / 64][MOM_VALUES];
#pragma unroll
    for (int q = 0; q < MOM_VALUES; ++q) {
        double v = acc[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < MOM_VALUES) {
        double v = 0.0;
        for (int w = 0; w < PRT_BLOCK / 64; ++w) v += sh[w][threadIdx.x];
        partials[(int64_t)blockIdx.x * MOM_VALUES + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(PRT_BLOCK) void k_moments_final(int nblocks_,
                                                             const double *__restrict__ partials,
                                                             double *__restrict__ out) {
    // fixed summation order: thread t adds blocks t, t+256, ...; then a fixed-shape tree
    __shared__ double sh[PRT_BLOCK][MOM_VALUES];
    double acc[MOM_VALUES] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks_; b += PRT_BLOCK)
#pragma unroll
        for (int q = 0; q < MOM_VALUES; ++q) acc[q] += partials[(int64_t)b * MOM_VALUES + q];
#pragma unroll
    for (int q = 0; q < MOM_VALUES; ++q) sh[threadIdx.x][q] = acc[q];
    __syncthreads();
    for (int stride = PRT_BLOCK / 2; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride)
#pragma unroll
            for (int q = 0; q < MOM_VALUES; ++q) sh[threadIdx.x][q] += sh[threadIdx.x + stride][q];
        __syncthreads();
    }
    if (threadIdx.x < MOM_VALUES) out[threadIdx.x] = sh[0][threadIdx.x];
}

// First stage for many rows (the fused march leaves one row per 512 rays): block g tree-reduces
// rows [g*256, (g+1)*256) in LDS (fixed shape) and writes row g of `stage`; k_moments_final then
// adds the few stage rows.
__global__ __launch_bounds__(PRT_BLOCK) void k_moments_stage(int nrows,
                                                             const double *__restrict__ partials,
                                                             double *__restrict__ stage) {
    __shared__ double sh[PRT_BLOCK][MOM_VALUES];
    const int row = blockIdx.x * PRT_BLOCK + threadIdx.x;
#pragma unroll
    for (int q = 0; q < MOM_VALUES; ++q)
        sh[threadIdx.x][q] = (row < nrows) ? partials[(int64_t)row * MOM_VALUES + q] : 0.0;
    __syncthreads();
    for (int stride = PRT_BLOCK / 2; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride)
#pragma unroll
            for (int q = 0; q < MOM_VALUES; ++q) sh[threadIdx.x][q] += sh[threadIdx.x + stride][q];
        __syncthreads();
    }
    if (threadIdx.x < MOM_VALUES) stage[(int64_t)blockIdx.x * MOM_VALUES + threadIdx.x] = sh[0][threadIdx.x];
}

// Order-preserving slot assignment inside one CMP_TILE (= 4 sub-tiles of PRT_BLOCK consecutive
// elements; thread t owns elements q*PRT_BLOCK + t, so loads and stores are lane-consecutive):
// ballot + popcount per wave, then a prefix over the 4 x 4 (sub-tile, wave) counts.
// keep[q] in -> pos[q] out (offset of the element among the tile's survivors).
PRT_DEV void tile_slots(const bool keep[CMP_ITEMS], int pos[CMP_ITEMS]) {
    __shared__ int cnt[CMP_ITEMS][PRT_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int rank[CMP_ITEMS];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        const unsigned long long b = __ballot(keep[q]);
        rank[q] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) cnt[q][wave] = __popcll(b);
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
#pragma unroll
        for (int w = 0; w < PRT_BLOCK / 64; ++w) {
            if (w == wave) pos[q] = run + rank[q];
            run += cnt[q][w];
        }
    }
}

// ---------------------------------------------------------------------------
// compaction: per-block popcount -> single-block scan of block totals -> scatter
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(PRT_BLOCK) void k_compact_count(const uint8_t *__restrict__ mask,
                                                             int64_t N,
                                                             int64_t *__restrict__ block_sums) {
    __shared__ int wsum[PRT_BLOCK / 64];
    const int64_t tile = (int64_t)blockIdx.x * CMP_TILE;
    int c = 0;
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        const int64_t idx = tile + q * PRT_BLOCK + threadIdx.x;
        if (idx < N && mask[idx]) ++c;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < PRT_BLOCK / 64; ++w) t += wsum[w];
        block_sums[blockIdx.x] = t;
    }
}

// exclusive scan of block_sums in place; total -> block_sums[nb]
__global__ __launch_bounds__(PRT_BLOCK) void k_compact_scan(int64_t *__restrict__ block_sums,
                                                            int64_t nb) {
    __shared__ int64_t part[PRT_BLOCK];
    const int64_t chunk = (nb + PRT_BLOCK - 1) / PRT_BLOCK;
    const int64_t lo = (int64_t)threadIdx.x * chunk;
    const int64_t hi = (lo + chunk < nb) ? lo + chunk : nb;
    int64_t s = 0;
    for (int64_t q = lo; q < hi; ++q) s += block_sums[q];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = 0;
        for (int q = 0; q < PRT_BLOCK; ++q) {
            const int64_t v = part[q];
            part[q] = run;
            run += v;
        }
        block_sums[nb] = run;
    }
    __syncthreads();
    int64_t run = part[threadIdx.x];
    for (int64_t q = lo; q < hi; ++q) {
        const int64_t v = block_sums[q];
        block_sums[q] = run;
        run += v;
    }
}

__global__ __launch_bounds__(PRT_BLOCK) void k_compact_scatter(
    const uint8_t *__restrict__ mask, int64_t N, const int64_t *__restrict__ block_offs,
    int32_t n_arrays, const double *const *__restrict__ src, double *const *__restrict__ dst,
    const int64_t *__restrict__ id_src, int64_t *__restrict__ id_dst,
    const uint8_t *__restrict__ u8_src, uint8_t *__restrict__ u8_dst) {
    const int64_t tile = (int64_t)blockIdx.x * CMP_TILE;
    bool keep[CMP_ITEMS];
    int pos[CMP_ITEMS];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        const int64_t idx = tile + q * PRT_BLOCK + threadIdx.x;
        keep[q] = (idx < N) && mask[idx];
    }
    tile_slots(keep, pos);
    const int64_t off = block_offs[blockIdx.x];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        if (!keep[q]) continue;
        const int64_t idx = tile + q * PRT_BLOCK + threadIdx.x;
        const int64_t o = off + pos[q];
        for (int a = 0; a < n_arrays; ++a) dst[a][o] = src[a][idx];
        if (id_src) id_dst[o] = id_src[idx];
        if (u8_src) u8_dst[o] = u8_src[idx];
    }
}

// ---------------------------------------------------------------------------
// device-side bundle generation: RectGrid.getGrid (sampling2d/raster.py:40-60) +
// OpticalSystemAnalysis.collimated_bundle (analysis/optical_system_analysis.py:83-122).
// The raster is reproduced BIT-EXACTLY: numpy.linspace is i*step + start with two
// roundings and the last sample forced to `stop`; the disk test is x*x + y*y <= 1 with
// separate roundings -- hence the explicit _rn intrinsics (no FMA contraction).
// ---------------------------------------------------------------------------
// a*b and a+b rounded separately: HIP's __dmul_rn/__dadd_rn are plain operators that hipcc
// contracts into FMAs, so contraction is switched off per statement instead
PRT_DEV double mul_rn(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
PRT_DEV double add_rn(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
PRT_DEV double lin_sample(int64_t i, int64_t n, double start, double step, double stop) {
    return (i == n - 1) ? stop : add_rn(mul_rn((double)i, step), start);
}

__global__ __launch_bounds__(PRT_BLOCK) void k_rectgrid_mask(int64_t n, double start, double step,
                                                             double stop,
                                                             uint8_t *__restrict__ mask) {
    const int64_t idx = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (idx >= n * n) return;
    const int64_t iy = idx / n, ix = idx - iy * n;  // np.meshgrid(x1d, x1d): x varies fastest
    const double x = lin_sample(ix, n, start, step, stop), y = lin_sample(iy, n, start, step, stop);
    mask[idx] = (add_rn(mul_rn(x, x), mul_rn(y, y)) <= 1.0) ? 1 : 0;
}

struct collimated_params {
    double radius, startx, starty, startz;
    double k[3], e[3];
};

__global__ __launch_bounds__(PRT_BLOCK) void k_rectgrid_scatter(
    const uint8_t *__restrict__ mask, int64_t n, double start, double step, double stop,
    const int64_t *__restrict__ block_offs, int64_t lo, int64_t hi, collimated_params prm,
    int64_t pitch, double *__restrict__ x_out, double *__restrict__ k_out,
    double *__restrict__ e_out) {
    const int64_t total = n * n;
    const int64_t tile = (int64_t)blockIdx.x * CMP_TILE;
    bool keep[CMP_ITEMS];
    int pos[CMP_ITEMS];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        const int64_t idx = tile + q * PRT_BLOCK + threadIdx.x;
        keep[q] = (idx < total) && mask[idx];
    }
    tile_slots(keep, pos);
    const int64_t off = block_offs[blockIdx.x];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        if (!keep[q]) continue;
        const int64_t p = off + pos[q];
        if (p < lo || p >= hi) continue;
        const int64_t idx = tile + q * PRT_BLOCK + threadIdx.x;
        const int64_t iy = idx / n, ix = idx - iy * n;
        const double px = lin_sample(ix, n, start, step, stop);
        const double py = lin_sample(iy, n, start, step, stop);
        const int64_t o = p - lo;
        // origin = radius * p + start (optical_system_analysis.py:106-108), two roundings
        x_out[o] = add_rn(mul_rn(prm.radius, px), prm.startx);
        x_out[pitch + o] = add_rn(mul_rn(prm.radius, py), prm.starty);
        x_out[2 * pitch + o] = prm.startz;
        if (k_out) {  // NULL: the caller keeps the bundle's k and E as one vector each (uniform first segment)
            k_out[o] = prm.k[0];
            k_out[pitch + o] = prm.k[1];
            k_out[2 * pitch + o] = prm.k[2];
        }
        if (e_out) {
            e_out[o] = prm.e[0];
            e_out[pitch + o] = prm.e[1];
            e_out[2 * pitch + o] = prm.e[2];
        }
    }
}

// ---------------------------------------------------------------------------
// Pupil rasters that are outer products of 1-d tables (sampling2d/raster.py:37-164) and the two
// bundle types built on them (analysis/optical_system_analysis.py:83-165).  Point (i, j) -- i slow,
// j fast, the order of np.meshgrid(...).reshape / .flatten -- is
//     px = xa[j] * xb[i],   py = ya[j] * yb[i]
// with the 1-d tables computed on the host exactly as the reference computes them (np.linspace,
// math.sin of the fan angle, ...), so that the samples are the reference's bit for bit: RectGrid and
// the two lattices of HexGrid (a table and a table of ones), CircularGrid (radii x cos / sin of the
// azimuths), Meridional / SagitalFan (a 1 x n raster).  clip: keep px*px + py*py <= 1, order kept.
// ---------------------------------------------------------------------------
struct raster_tables {
    const double *xa, *xb, *ya, *yb;  // device copies: xa, ya have nj entries, xb, yb ni
    int64_t ni, nj;
};

PRT_DEV void raster_point(const raster_tables &t, int64_t idx, double &px, double &py) {
    const int64_t i = idx / t.nj, j = idx - i * t.nj;
    px = mul_rn(t.xa[j], t.xb[i]);
    py = mul_rn(t.ya[j], t.yb[i]);
}

__global__ __launch_bounds__(PRT_BLOCK) void k_raster_mask(raster_tables t, int32_t clip,
                                                           uint8_t *__restrict__ mask) {
    const int64_t idx = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (idx >= t.ni * t.nj) return;
    double px, py;
    raster_point(t, idx, px, py);
    mask[idx] = (!clip || add_rn(mul_rn(px, px), mul_rn(py, py)) <= 1.0) ? 1 : 0;
}

struct bundle_params {
    int32_t kind;  // 0 collimated, 1 divergent
    double radius, startx, starty, startz, anglex, angley, index;
    double k[3], e[3];  // collimated: the bundle's wave vector and E field
};

__global__ __launch_bounds__(PRT_BLOCK) void k_raster_bundle(
    const uint8_t *__restrict__ mask, raster_tables t, const int64_t *__restrict__ block_offs, int64_t lo,
    int64_t hi, bundle_params prm, int64_t pitch, double *__restrict__ x_out, double *__restrict__ k_out,
    double *__restrict__ e_out, double *__restrict__ p_out) {
    const int64_t total = t.ni * t.nj;
    const int64_t tile = (int64_t)blockIdx.x * CMP_TILE;
    bool keep[CMP_ITEMS];
    int pos[CMP_ITEMS];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        const int64_t idx = tile + q * PRT_BLOCK + threadIdx.x;
        keep[q] = (idx < total) && mask[idx];
    }
    tile_slots(keep, pos);
    const int64_t off = block_offs[blockIdx.x];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        if (!keep[q]) continue;
        const int64_t p = off + pos[q];
        if (p < lo || p >= hi) continue;
        double px, py;
        raster_point(t, tile + q * PRT_BLOCK + threadIdx.x, px, py);
        const int64_t o = p - lo;
        if (p_out) {
            p_out[o] = px;
            p_out[pitch + o] = py;
        }
        vec3 k, e;
        if (prm.kind == 0) {
            // origin = radius * p + start (optical_system_analysis.py:106-108), two roundings
            x_out[o] = add_rn(mul_rn(prm.radius, px), prm.startx);
            x_out[pitch + o] = add_rn(mul_rn(prm.radius, py), prm.starty);
            x_out[2 * pitch + o] = prm.startz;
            k = v3(prm.k[0], prm.k[1], prm.k[2]);
            e = v3(prm.e[0], prm.e[1], prm.e[2]);
        } else {
            // every ray starts at the source point; directions fan out over the pupil angles
            // (optical_system_analysis.py:150-158); k = n * unit vector
            x_out[o] = prm.startx;
            x_out[pitch + o] = prm.starty;
            x_out[2 * pitch + o] = prm.startz;
            const double ay = add_rn(prm.angley, mul_rn(prm.radius, px));
            const double ax = add_rn(prm.anglex, mul_rn(prm.radius, py));
            const double cax = cos(ax);
            k = v3(mul_rn(prm.index, mul_rn(sin(ay), cax)), mul_rn(prm.index, sin(ax)),
                   mul_rn(prm.index, mul_rn(cos(ay), cax)));
            // E: the unit vector perpendicular to k that prt_efield_perp picks
            const double fx = fabs(k.x), fy = fabs(k.y), fz = fabs(k.z);
            const vec3 a = (fy <= fx && fy <= fz) ? v3(0, 1, 0) : ((fx <= fz) ? v3(1, 0, 0) : v3(0, 0, 1));
            e = normalized(cross(k, a));
        }
        if (k_out) {
            k_out[o] = k.x;
            k_out[pitch + o] = k.y;
            k_out[2 * pitch + o] = k.z;
        }
        if (e_out) {
            e_out[o] = e.x;
            e_out[pitch + o] = e.y;
            e_out[2 * pitch + o] = e.z;
        }
    }
}
