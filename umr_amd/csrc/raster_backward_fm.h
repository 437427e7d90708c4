// raster_backward_fm.h -- BODY of the face-major backward kernel; NOT a standalone header: raster_backward.h includes it once
// per (hand-out form, register budget, gradient routing) as separate kernels, because the forms want different register
// allocations (merging two forms behind `if constexpr` in ONE function once cost the vertex + texel variant 10 % on its
// untouched path):
//   FM_QUADS 0  k_raster_backward_fm      4x4 sub-tiles handed out by v_readlane rounds, four per visit
//   FM_QUADS 0  k_raster_backward_fm_w6   the same at 6 waves / SIMD (vertex + texel variant: no SGPR spills in the visit, -5 %)
//   FM_QUADS 1  k_raster_backward_fm_quads   2x2 quads through an LDS list, sixteen per visit (silhouette variant: -14 %)
//   FM_QUADS 1, FM_ALPHA_GEOM 1  k_raster_backward_fm_ag   one pass for a render whose alpha gradient goes to the geometry and
//               whose rgb gradient goes to the texels only (UMR_BWD_ALPHA_GEOMETRY)
//   ... + FM_PACKED 1  k_raster_backward_fm_agp   the same reading the forward's PACKED saved state (RasterArgs::state: one
//               256-byte record per 4x4 tile with per-quad cull summaries) instead of the planes (UMR_BWD_PACKED_STATE)
// Why quads (tools/visit_census.py: this source with counters on the emulator): of the lanes a 4x4 hand-out carries 66 % lie
// inside the face's band; by 2x2 pieces it is 89 %.  Wave visits per mesh of the SURVEY 8d scene: silhouette 12 694 -> 8 878,
// texel-only 14 266 -> 12 140, vertex + texel 17 520 -> 14 956.  Measured on MI355X (us, N = 16): silhouette (N = 32) 158 -> 136,
// texel-only 134 -> 126.5 (only with FM_VREC 0: at the 72-VGPR budget its allocation swings between 126 and 141 with
// incidental source changes), vertex + texel 213 -> 202 -- so only the silhouette variant and the one-pass kernel take it.
// Work items (round 6).  With A.order set (k_face_order, raster_backward.h) a wave does not own a face but an ITEM of the launch's
// list: a whole face, or -- for a face whose estimated work exceeds the split threshold -- part p of its n parts, a contiguous share
// of the face's culling passes.  A part leaves its partial sums in a slab and k_split_reduce, the next launch, adds the parts up in
// part order: on the geometry a training step really renders (profiles/scenes/) a handful of faces carry 20x the median work
// and one wave per face ran for the whole launch (one-pass backward 268 / 207 us -> 107 / 109 us on the two frozen scenes,
// unchanged 136 us on the regular SURVEY 8d scene).  The unsplit path is the old one, instruction for instruction in the visit.
#ifndef UMR_MUL24
#ifdef UMR_HOST_SHIM
#define UMR_MUL24(a, b) ((a) * (b))
#else
#define UMR_MUL24(a, b) __mul24(a, b)     // v_mul_u32_u24 / v_mad_u32_u24: full rate (v_mul_lo_u32 is quarter rate)
#endif
#endif
template <int RGB, bool NEED_GF, bool NEED_GT, bool COMMON>  // RGB 2 = silhouette only (soft_colors / grads are alpha planes)
// COMMON = the production case (gradient arrives 2x2-pooled, power-of-two image, double-sided faces) as compile-time
// facts: the wave-uniform flags otherwise live as 64-bit lane masks in SGPRs that spill (v_readlane per visit)
#ifndef BWD_WPE
#define BWD_WPE 7
#endif
#ifndef BWD_WPE_ATTR
#define BWD_WPE_ATTR __attribute__((amdgpu_waves_per_eu(BWD_WPE, BWD_WPE)))
#endif
__global__ __launch_bounds__(FM_WAVES * 64) BWD_WPE_ATTR void FM_KERNEL_NAME(const RasterArgs A) {
#ifdef UMR_HOST_SHIM   // tests/host_kernel/wave_emu.h: the launch's dynamic LDS
    float *s_tex = (float *)umr_host_dynamic_lds();
#else
    extern __shared__ __attribute__((aligned(16))) float s_tex[];  // [FM_WAVES][FM_TEXCOPY][FM_TEX_STRIDE(TS)]
#endif
#if FM_QUADS
    // quad hand-out: the culling lanes refine their surviving 4x4 sub-tile into its four 2x2 quads (band test + saved-state test
    // per quad) and write one packed origin (qx | qy << 16, in quad units) per surviving quad at its rank; a visit hands 16 quads
    // to the 16 lane groups of 4
    constexpr bool PK = FM_PACKED != 0;               // packed saved state: a slot also carries the quad's byte offset in it
    constexpr bool PK2 = PK && !(FM_AGP_SLOT16 != 0 && COMMON);   // ... as the second word of an 8-byte slot
    __shared__ unsigned s_quad[FM_WAVES][(PK || (RGB == 2 && COMMON)) ? 1 : 256 + 32];   // (+32: the prefetch of the visit after the last reads past the end)
    __shared__ uint2 s_quad2[FM_WAVES][PK2 ? 256 + 32 : 1];
    // silhouette variant on a power-of-two image: the slot holds the quad origin's pixel-centre coordinates (exact floats) and its
    // byte offsets into the full / pooled planes, as the 4x4 slots form does -- the visit adds its lane's place, no integer decode
    // ... the packed-state one-pass kernel too (FM_AGP_SLOT16): .z is then the quad's byte offset in the packed state
    constexpr bool QSLOT16 = (RGB == 2 || (FM_PACKED != 0 && FM_AGP_SLOT16 != 0)) && COMMON;
    __shared__ float4 s_quad4[FM_WAVES][QSLOT16 ? 256 + 32 : 1];
#endif
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: uniform
    const int F = A.F, IS = A.IS, TS = A.TS;
    // AG: the rgb gradient reaches the texels only and the alpha gradient the geometry (silhouette backward + texel-only
    // backward of one render in one pass); instantiated as <1, true, true, .>
    constexpr bool AG = FM_ALPHA_GEOM != 0;
    const bool pooled = COMMON ? true : (A.grad_pooled != 0);
    const bool two_sided = COMMON ? true : (A.double_side != 0);
    // Wave-uniform float constants of the visit body, held in VGPRs on purpose: with the 32-float face record in SGPRs
    // the scalar file is full, and every constant the allocator spills comes back as a v_readlane (VALU) per visit.
    // As VALU operands they are as cheap from a VGPR.  (#define FM_VCONST 0 keeps them scalar for A/B.)
#ifndef FM_V
#if FM_VCONST && !defined(UMR_HOST_SHIM)
#define FM_V(x) ({ float v_; asm volatile("v_mov_b32 %0, %1" : "=v"(v_) : "s"(x)); v_; })
#else
#define FM_V(x) (x)
#endif
#endif
    const float c_near = FM_V(A.near_), c_far = FM_V(A.far_), c_rr = FM_V(A.r_range), c_ig = FM_V(A.inv_gamma);
    const float c_thr2 = FM_V(A.threshold), c_nis = FM_V(A.nis);
    // XCD-aware: hardware XCD = blockIdx % 8.  Each XCD owns a fixed contiguous EIGHTH of every mesh's faces
    // (index-neighbouring faces of a subdivided mesh are spatial neighbours), so the per-pixel state its waves
    // re-read (~6x) covers 1/8 of the screen and stays in that XCD's 4 MB L2, and all 8 XCDs share every mesh
    // (balance at small N).  Measured fabric reads: 47 MB/mesh round-robin -> ~20 MB/mesh (11.8 MB algorithmic).
    const int fblocks = (F + FM_WAVES - 1) / FM_WAVES;   // blocks per mesh (grid = N * fblocks)
    int nb = blockIdx.x / fblocks, fb = blockIdx.x % fblocks;
    if (fblocks % 8 == 0 && (A.N * fblocks) % 8 == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = fblocks >> 3;
        nb = slot / per;
        fb = __builtin_amdgcn_readfirstlane(fm_owned_face(xcd, slot % per, per, A.fm_split));   // (uniform; the division hides it)
    }
    int fidx = fb * FM_WAVES + wave;
    // this wave's work item (k_face_order): a whole face, or part `part` of the `nparts` a heavy face was split into.  Only the
    // item word itself stays live across the walk (the epilogue decodes it again): every wave-uniform value more is an SGPR
    // spill in the visit loop
    unsigned item = 0u;          // part << 26 | (nparts - 1) << 21 | ...
    if (FM_WAVES == 1 && A.order) {   // work items in cost order: same XCD ownership, heavy items first
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int g = slot / A.order_stride;
        const unsigned ex = A.order[((size_t)g * 8 + xcd) * A.order_stride + slot % A.order_stride].x;
        if ((int)ex < 0) return;      // padding of the list (whole workgroup = this wave)
        nb = g * A.order_group + (int)((ex >> 16) & 31u);
        fidx = (int)(ex & 0xffffu);
        item = (unsigned)__builtin_amdgcn_readfirstlane((int)ex);
    }
    const bool live = fidx < F;
    // (explicitly wave-uniform: the record's address below feeds scalar loads)
    const int n = __builtin_amdgcn_readfirstlane(nb), f = __builtin_amdgcn_readfirstlane(live ? fidx : 0);
    const size_t npix = (size_t)IS * IS;
    // wave-uniform bases of this mesh's per-pixel planes; every per-pixel load below is base + 32-bit byte offset
    const int H2 = IS >> 1;
    const unsigned pst = (unsigned)(npix * sizeof(float));                      // plane stride in bytes
    const unsigned gps = pooled ? (unsigned)((size_t)H2 * H2 * sizeof(float)) : pst;
    const int cplanes = RGB == 2 ? 1 : 4;
    const char *sc_n = (const char *)(A.soft_colors + (size_t)n * cplanes * npix);
    const char *ag_n = (const char *)(A.aggrs + (size_t)n * 2 * npix);
    const char *gc_n = (const char *)(A.grad_colors + (size_t)n * cplanes * (pooled ? (size_t)H2 * H2 : npix));
#if FM_PACKED
    const char *st_n = (const char *)(A.state + (size_t)n * npix * (STATE_REC / 16));   // 16 B of state per pixel
    (void)ag_n; (void)sc_n;
    const unsigned tpr = (unsigned)IS >> 2;                                            // tile records per row
#endif
    float *wave_tex = s_tex + (size_t)wave * FM_TEXCOPY * FM_TEX_STRIDE(TS);
    // this lane's copy: horizontally and vertically adjacent pixels of a 4x4 / 8x8 tile get different copies
    float *my_tex = wave_tex + ((lane ^ (lane >> 2) ^ (lane >> 4)) & (FM_TEXCOPY - 1)) * FM_TEX_STRIDE(TS);
    if (NEED_GT && TS > 1)
        for (int j = lane; j < FM_TEXCOPY * FM_TEX_STRIDE(TS); j += 64) wave_tex[j] = 0.f;
    float gv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gt0 = 0.f, gt1 = 0.f, gt2 = 0.f;  // TS == 1 texel gradient
    bool visited = false;                   // wave-uniform: some sub-tile survived the culling pass
    if (live) {
        // VGPR-resident operands where the register budget of 7 waves / SIMD has room for them (silhouette and
        // texel-gradient-only variants; the full variant would spill)
        constexpr bool VREC = FM_VREC != 0 && (RGB == 2 || !NEED_GF || (FM_ALPHA_GEOM != 0 && FM_AG_VREC != 0));   // (the one-pass kernel: 21 VGPR spills with the copies at 7 waves)
        typename std::conditional<VREC, FaceV, Face>::type fc;
        load_face(fc, A.rec + ((size_t)n * F + f) * REC);
        if constexpr (VREC) fc.fill();
        const float *__restrict__ tex_f = A.textures + ((size_t)(n / A.tex_group) * F + f) * TS * 3;
        // pixel-index window of the dilated bbox, widened by one pixel; the exact per-pixel reject of the
        // reference (:536) still runs inside eval_pair, so the window only has to be conservative.
        // xp(i) = (2i + 1 - IS)/IS  <=>  i = (xp*IS + IS - 1)/2
        const float h = 0.5f * IS;
        int x0 = (int)floorf(fc.template g<R_XLO>() * h + h - 0.5f) - 1, x1 = (int)ceilf(fc.template g<R_XHI>() * h + h - 0.5f) + 1;
        int yi0 = (int)floorf(fc.template g<R_YLO>() * h + h - 0.5f) - 1, yi1 = (int)ceilf(fc.template g<R_YHI>() * h + h - 0.5f) + 1;
        // NaN / inf bounds: comparisons below fail safe to the full image (the reference would visit all pixels)
        if (!(fc.template g<R_XLO>() == fc.template g<R_XLO>() && fc.template g<R_XHI>() == fc.template g<R_XHI>() && fc.template g<R_YLO>() == fc.template g<R_YLO>() && fc.template g<R_YHI>() == fc.template g<R_YHI>())) { x0 = 0; x1 = IS - 1; yi0 = 0; yi1 = IS - 1; }
        x0 = max(x0, 0); x1 = min(x1, IS - 1); yi0 = max(yi0, 0); yi1 = min(yi1, IS - 1);
        const int r0 = IS - 1 - yi1, r1 = IS - 1 - yi0;  // row = IS-1-yi
        if (x0 <= x1 && r0 <= r1) {
            // Sub-tiles of FM_TW x FM_TH pixels, FM_NQ = 64 / (FM_TW * FM_TH) of them per wave visit: the face is
            // wave-uniform here, so the 64 lanes need not form ONE tile -- each group of FM_TW*FM_TH lanes takes its own
            // needed sub-tile of this face.  4x4 sub-tiles fill 74 % of their lanes with contributing pixels against
            // 54 % for one 8x8 tile (CPU simulation of the culling, 1280-face sphere at IS = 512).
            const int tx0 = x0 / FM_TW, tx1 = x1 / FM_TW, ty0 = r0 / FM_TH, ty1 = r1 / FM_TH;
            const bool pow2 = COMMON ? true : ((IS & (IS - 1)) == 0);
            const float inv_is = 1.f / (float)IS;
            const int ntx = tx1 - tx0 + 1, ntiles = ntx * (ty1 - ty0 + 1);
            // the record's floats [R_I0, R_I0 + 12) as tile_may_hit takes them
            const float4 i0 = make_float4(fc.template g<R_I0>(), fc.template g<R_I3>(), fc.template g<R_I1>(), fc.template g<R_I4>());
            const float4 i1 = make_float4(fc.template g<R_I2>(), fc.template g<R_I5>(), fc.template g<R_I6>(), fc.template g<R_I7>());
            const float4 i2 = make_float4(fc.template g<R_I8>(), fc.template g<R_K2>(), fc.template g<R_K0>(), fc.template g<R_K1>());
            const float thr_cull = A.thr + A.rec[((size_t)n * F + f) * REC + R_CULL];   // band of the cull: + the reference's noise
            const int sub = lane / (FM_TW * FM_TH), sl = lane % (FM_TW * FM_TH);   // sub-tile slot of this lane, lane in it
            (void)sub; (void)sl;
#if FM_QUADS
            const int qsub = lane >> 2, qlx = lane & 1, qly = (lane >> 1) & 1;
            const float qlxf = (float)(2 * qlx) * inv_is, qlyf = (float)(2 * qly) * inv_is;
            const unsigned qlo_pn = (unsigned)(qly * IS + qlx) * 4u;
            (void)qlo_pn;
#if FM_PACKED
            const unsigned qlo_st = (unsigned)(qly * 4 + qlx) * 4u;      // this lane's place in its quad, in a tile record
#endif
#endif
            // this item's culling passes: a contiguous share of the face's ceil(ntiles / 64) (all of them for an unsplit face)
            int tb_begin = 0, tb_end = ntiles;
            if (item >> 21) {
                const int nparts = (int)((item >> 21) & 31u) + 1, part = (int)(item >> 26);
                const int ppp = (((ntiles + 63) >> 6) + nparts - 1) / nparts;
                tb_begin = min(part * ppp * 64, ntiles);
                tb_end = min(tb_begin + ppp * 64, ntiles);
            }
#ifdef FM_NO_CULL            // time-split experiment (-DFM_NO_CULL, HISTORY.md 4.2): per-face set-up and reductions only
            for (int tb = tb_end; tb < tb_end; tb += 64) {
#else
            for (int tb = tb_begin; tb < tb_end; tb += 64) {
#endif
                // one lane per sub-tile: drop those no pixel of which can survive (conservative), then walk the rest
                const int ti = tb + lane;
                bool want = false;
                int tpk = 0;   // packed (tx, ty) of this lane's candidate
#if FM_QUADS
                unsigned qm = 0;   // surviving 2x2 quads of the candidate: bit q = quad (q & 1, q >> 1)
#endif
                if (ti < tb_end) {
                    const int ttx = tx0 + ti % ntx, tty = ty0 + ti / ntx;
                    tpk = ttx | (tty << 16);
                    const int px0 = ttx * FM_TW, px1 = min(px0 + FM_TW - 1, IS - 1), pr0 = tty * FM_TH, pr1 = min(pr0 + FM_TH - 1, IS - 1);
                    const float cxl = ndc_coord_fast(px0, IS, inv_is, pow2), cxh = ndc_coord_fast(px1, IS, inv_is, pow2);
                    const float cyh = ndc_coord_fast(IS - 1 - pr0, IS, inv_is, pow2), cyl = ndc_coord_fast(IS - 1 - pr1, IS, inv_is, pow2);
#if FM_QUADS
                    const bool whole = pow2 && IS >= 4 && FM_TW == 4 && FM_TH == 4;   // no ragged sub-tile: one evaluation for the sub-tile and its quads
                    if (whole) {
                        // (loop-variant VGPR copies: the per-face products of the quad test are then recomputed per pass -- a dozen
                        // instructions -- instead of living in SGPRs across the visit loop, where they spill)
                        float pxv = 2.f * inv_is, thrv = thr_cull;
#ifndef UMR_HOST_SHIM
                        asm volatile("" : "+v"(pxv), "+v"(thrv));
#endif
                        qm = subtile_quads_may_hit(i0, i1, i2, 0.5f * (cxl + cxh), 0.5f * (cyl + cyh), pxv, thrv);
                        want = qm != 0;
                    } else
#endif
                    want = tile_may_hit(i0, i1, i2, 0.5f * (cxl + cxh), 0.5f * (cyl + cyh), 0.5f * (cxh - cxl),
                                        0.5f * (cyh - cyl), thr_cull);
#if FM_QUADS
                    if (want && !whole) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int qx0 = px0 + 2 * (q & 1), qr0 = pr0 + 2 * (q >> 1);
                            if (qx0 < IS && qr0 < IS) {
                                const int qx1 = min(qx0 + 1, IS - 1), qr1 = min(qr0 + 1, IS - 1);
                                const float axl = ndc_coord_fast(qx0, IS, inv_is, pow2), axh = ndc_coord_fast(qx1, IS, inv_is, pow2);
                                const float ayh = ndc_coord_fast(IS - 1 - qr0, IS, inv_is, pow2), ayl = ndc_coord_fast(IS - 1 - qr1, IS, inv_is, pow2);
                                if (tile_may_hit(i0, i1, i2, 0.5f * (axl + axh), 0.5f * (ayl + ayh), 0.5f * (axh - axl), 0.5f * (ayh - ayl), thr_cull))
                                    qm |= 1u << q;
                            }
                        }
                        want = qm != 0;
                    }
#endif
#if FM_STATE_CULL
                    // Exact sub-tile skips from the saved forward state, decided HERE by the one lane that owns the
                    // candidate (64 candidates per pass) instead of by a whole wave visit that finds its four sub-tiles dead:
                    //  * silhouette: every pixel of the sub-tile has alpha == 1.0f -> g (1 - alpha) finite = 0 (:584);
                    //  * texel gradients only, soft-max: even the face's nearest depth is >= 89 gamma behind the soft-max
                    //    maximum of every pixel -> p = D exp(<-89) / S = 0.0f (:608); hard mode: the face wins no pixel (:596).
                    // Inside the silhouette that removes most of the back-facing half of the mesh before any visit.
#if FM_PACKED
                    if (want) {   // (IS is a multiple of 8: no ragged sub-tile) the record's quad summaries decide, 32 B per candidate
                        const unsigned to = (UMR_MUL24((unsigned)pr0 >> 2, tpr) + ((unsigned)px0 >> 2)) * (STATE_REC * 4u);
                        const float4 mn4 = ld_u4(st_n, to + STATE_O_QMIN * 4u), op4 = ld_u4(st_n, to + STATE_O_QOPAQUE * 4u);
                        const float zmin_c = fminf(fminf(fc.template g<R_Z0>(), fc.template g<R_Z1>()), fc.template g<R_Z2>());
                        const float zq = (c_far - zmin_c) * c_rr;
                        // a quad dies only when BOTH terms vanish: depth-dead (NaN maximum: compares false, visited) AND alpha == 1
                        unsigned alive = 0;
                        alive |= (!((zq - mn4.x) * c_ig < -89.f) || !(op4.x == 1.f)) ? 1u : 0u;
                        alive |= (!((zq - mn4.y) * c_ig < -89.f) || !(op4.y == 1.f)) ? 2u : 0u;
                        alive |= (!((zq - mn4.z) * c_ig < -89.f) || !(op4.z == 1.f)) ? 4u : 0u;
                        alive |= (!((zq - mn4.w) * c_ig < -89.f) || !(op4.w == 1.f)) ? 8u : 0u;
                        qm &= alive;
                        want = qm != 0;
                    }
#else
                    if (FM_TW == 4 && FM_TH == 4 && (RGB == 2 || !NEED_GF || (AG && FM_QUADS)) && want && px0 + 3 < IS && pr0 + 3 < IS && (IS & 3) == 0) {
                        const char *plane = RGB == 2 ? sc_n : ag_n + pst;           // alpha | soft-max maximum (hard: face id)
                        const unsigned o0 = (unsigned)(pr0 * IS + px0) * 4u, rs = (unsigned)IS * 4u;
                        const float4 q0 = ld_u4(plane, o0), q1 = ld_u4(plane, o0 + rs), q2 = ld_u4(plane, o0 + 2u * rs),
                                     q3 = ld_u4(plane, o0 + 3u * rs);
#if FM_QUADS
                        // the same exact skips per QUAD: quad (0,0) = rows 0-1 x columns 0-1, (1,0) = columns 2-3, (.,1) = rows 2-3.  NaN in
                        // the saved state: never skipped (compares false / the sum test), as the per-pixel test of the visit has it.
                        // One-pass kernel: a quad dies only when BOTH terms vanish (depth-dead AND alpha == 1)
                        const float e[4][4] = {{q0.x, q0.y, q1.x, q1.y}, {q0.z, q0.w, q1.z, q1.w}, {q2.x, q2.y, q3.x, q3.y}, {q2.z, q2.w, q3.z, q3.w}};
                        unsigned alive = 0;
                        float ea[4][4] = {};
                        if (AG) {   // second plane: the render's alpha (the geometry term dies where it is 1.0f)
                            const char *pa = sc_n + 3u * pst;
                            const float4 a0 = ld_u4(pa, o0), a1 = ld_u4(pa, o0 + rs), a2 = ld_u4(pa, o0 + 2u * rs), a3 = ld_u4(pa, o0 + 3u * rs);
                            const float t[4][4] = {{a0.x, a0.y, a1.x, a1.y}, {a0.z, a0.w, a1.z, a1.w}, {a2.x, a2.y, a3.x, a3.y}, {a2.z, a2.w, a3.z, a3.w}};
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int k = 0; k < 4; ++k) ea[q][k] = t[q][k];
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            bool al;
                            if (RGB == 2) {
                                al = !((e[q][0] == 1.f) & (e[q][1] == 1.f) & (e[q][2] == 1.f) & (e[q][3] == 1.f));
                            } else if (RGB == 1) {
                                const float mn = fminf(fminf(e[q][0], e[q][1]), fminf(e[q][2], e[q][3]));
                                const float zmin_c = fminf(fminf(fc.template g<R_Z0>(), fc.template g<R_Z1>()), fc.template g<R_Z2>());
                                const float sm = (e[q][0] + e[q][1]) + (e[q][2] + e[q][3]);
                                al = !(((c_far - zmin_c) * c_rr - mn) * c_ig < -89.f) || !(sm == sm);
                                if (AG) al = al || !((ea[q][0] == 1.f) & (ea[q][1] == 1.f) & (ea[q][2] == 1.f) & (ea[q][3] == 1.f));
                            } else {
                                const float ff = (float)f;
                                al = (e[q][0] == ff) | (e[q][1] == ff) | (e[q][2] == ff) | (e[q][3] == ff);
                            }
                            alive |= al ? 1u << q : 0u;
                        }
                        qm &= alive;
                        want = qm != 0;
#else
                        if (RGB == 2) {
                            want = !((q0.x == 1.f) & (q0.y == 1.f) & (q0.z == 1.f) & (q0.w == 1.f) & (q1.x == 1.f) & (q1.y == 1.f) &
                                     (q1.z == 1.f) & (q1.w == 1.f) & (q2.x == 1.f) & (q2.y == 1.f) & (q2.z == 1.f) & (q2.w == 1.f) &
                                     (q3.x == 1.f) & (q3.y == 1.f) & (q3.z == 1.f) & (q3.w == 1.f));
                        } else if (RGB == 1) {
                            const float mn = fminf(fminf(fminf(fminf(q0.x, q0.y), fminf(q0.z, q0.w)), fminf(fminf(q1.x, q1.y), fminf(q1.z, q1.w))),
                                                   fminf(fminf(fminf(q2.x, q2.y), fminf(q2.z, q2.w)), fminf(fminf(q3.x, q3.y), fminf(q3.z, q3.w))));
                            const float zmin_c = fminf(fminf(fc.template g<R_Z0>(), fc.template g<R_Z1>()), fc.template g<R_Z2>());
                            // same expression as the per-pixel test below; monotone in the maximum, so the sub-tile's smallest
                            // maximum decides for all 16 pixels.  fminf drops NaN operands, so a NaN in the saved state is
                            // looked for explicitly (the sum of the 16 values is NaN iff one of them is, or +inf - inf):
                            // such a sub-tile is visited, as the per-pixel test below would have it
                            const float sm = ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w)) +
                                             (((q2.x + q2.y) + (q2.z + q2.w)) + ((q3.x + q3.y) + (q3.z + q3.w)));
                            want = !(((c_far - zmin_c) * c_rr - mn) * c_ig < -89.f) || !(sm == sm);
                        } else {
                            const float ff = (float)f;
                            want = (q0.x == ff) | (q0.y == ff) | (q0.z == ff) | (q0.w == ff) | (q1.x == ff) | (q1.y == ff) | (q1.z == ff) |
                                   (q1.w == ff) | (q2.x == ff) | (q2.y == ff) | (q2.z == ff) | (q2.w == ff) | (q3.x == ff) | (q3.y == ff) |
                                   (q3.z == ff) | (q3.w == ff);
                        }
#endif
                    }
#endif
#endif
                }
                unsigned long long tm = __ballot(want);
                visited |= tm != 0;
#ifdef FM_NO_VISIT          // time-split experiment (-DFM_NO_VISIT, HISTORY.md 4.2): per-face set-up + culling pass only
                tm = 0;
#endif
#if FM_QUADS
                // this lane's first quad's rank among the wave's surviving quads, in (lane, quad) order: prefix sum of the lanes' quad
                // counts (0 .. 4) from three ballots of the count's bits
                const int cnt = tm != 0 ? __popcll((unsigned long long)qm) : 0;
                const unsigned long long b0 = __ballot((cnt & 1) != 0), b1 = __ballot((cnt & 2) != 0), b2 = __ballot((cnt & 4) != 0);
                const int nq = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
                if (want) {
                    int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b0, 0u)) +
                              2 * (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0u)) +
                              4 * (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0u));
                    const int px0 = (tpk & 0xffff) * FM_TW, pr0 = (tpk >> 16) * FM_TH;
                    if constexpr (QSLOT16) {
                        const float x0 = ndc_coord_fast(px0, IS, inv_is, true), y0 = ndc_coord_fast(IS - 1 - pr0, IS, inv_is, true);
                        const float dq = 4.f * inv_is;                                        // two pixels
                        const unsigned o_gp = (unsigned)((pr0 >> 1) * H2 + (px0 >> 1)) * 4u, r_gp = (unsigned)H2 * 4u;   // (one pooled row down)
#if FM_PACKED
                        // packed state: the sub-tile IS one record; quad (qx, qy) starts 8 qx + 32 qy bytes into each of its planes
                        const unsigned o_pn = (UMR_MUL24((unsigned)pr0 >> 2, tpr) + ((unsigned)px0 >> 2)) * (STATE_REC * 4u), r_pn = 32u;
#else
                        const unsigned o_pn = (unsigned)(pr0 * IS + px0) * 4u, r_pn = 2u * (unsigned)IS * 4u;             // two rows down
#endif
                        if (qm & 1u) s_quad4[wave][pos++] = make_float4(x0, y0, __int_as_float((int)o_pn), __int_as_float((int)o_gp));
                        if (qm & 2u) s_quad4[wave][pos++] = make_float4(x0 + dq, y0, __int_as_float((int)(o_pn + 8u)), __int_as_float((int)(o_gp + 4u)));
                        if (qm & 4u) s_quad4[wave][pos++] = make_float4(x0, y0 - dq, __int_as_float((int)(o_pn + r_pn)), __int_as_float((int)(o_gp + r_gp)));
                        if (qm & 8u) s_quad4[wave][pos++] = make_float4(x0 + dq, y0 - dq, __int_as_float((int)(o_pn + r_pn + 8u)), __int_as_float((int)(o_gp + r_gp + 4u)));
                    } else {
                    // origin of a quad in PIXELS: x | row << 16 (both even: a lane ORs its place in the quad in)
                    const unsigned base = (unsigned)px0 | ((unsigned)pr0 << 16);
#if FM_PACKED
                    // ... and its byte offset in the mesh's packed state: quad (qx, qy) of a tile starts 8 qx + 32 qy bytes into a plane
                    const unsigned to = (UMR_MUL24((unsigned)pr0 >> 2, tpr) + ((unsigned)px0 >> 2)) * (STATE_REC * 4u);
                    if (qm & 1u) s_quad2[wave][pos++] = make_uint2(base, to);
                    if (qm & 2u) s_quad2[wave][pos++] = make_uint2(base + 2u, to + 8u);
                    if (qm & 4u) s_quad2[wave][pos++] = make_uint2(base + 0x20000u, to + 32u);
                    if (qm & 8u) s_quad2[wave][pos++] = make_uint2(base + 0x20002u, to + 40u);
#else
                    if (qm & 1u) s_quad[wave][pos++] = base;
                    if (qm & 2u) s_quad[wave][pos++] = base + 2u;
                    if (qm & 4u) s_quad[wave][pos++] = base + 0x20000u;
                    if (qm & 8u) s_quad[wave][pos++] = base + 0x20002u;
#endif
                    }
                }
                UMR_WAVE_LDS_HANDOVER();
                const unsigned *qslot = &s_quad[wave][(PK || QSLOT16) ? 0 : qsub];
                const uint2 *qslot2 = &s_quad2[wave][PK2 ? qsub : 0];
                const float4 *qslot4 = &s_quad4[wave][QSLOT16 ? qsub : 0];
                unsigned qe_next = (QSLOT16 || PK) ? 0u : qslot[0];
                uint2 q2_next = PK2 ? qslot2[0] : make_uint2(0u, 0u);
                float4 q4_next = QSLOT16 ? qslot4[0] : make_float4(0.f, 0.f, 0.f, 0.f);
                for (int v0 = 0; v0 < nq; v0 += 16) {
                    const int mine = v0 + qsub < nq ? 0 : -1;
                    const unsigned qe = PK2 ? q2_next.x : qe_next;
                    const unsigned qst = q2_next.y;
                    const float4 q4 = q4_next;
                    if constexpr (QSLOT16) q4_next = qslot4[v0 + 16];
                    else if constexpr (PK2) q2_next = qslot2[v0 + 16];
                    else qe_next = qslot[v0 + 16];     // (past the last quad: stale or unwritten words of the array, never used)
                    (void)qst;
#else
                while (tm) {
                    // next FM_NQ wanted sub-tiles, one per lane group (groups past the last one idle this visit)
                    int mine = -1;
#pragma unroll
                    for (int qq = 0; qq < FM_NQ; ++qq) {
                        if (tm) {
                            const int tbit = __builtin_ctzll(tm);
                            tm &= tm - 1;
                            const int e = __builtin_amdgcn_readlane(tpk, tbit);
                            if (sub == qq) mine = e;
                        }
                    }
#endif
                    if (FM_RELOAD_PER_TILE && NEED_GF && RGB != 2 && !AG) {
                        // re-fetch the record from the scalar cache every visit: keeps the 32 constants loop-VARIANT so the
                        // compiler cannot hoist 30+ SGPR->VGPR copies out of the tile loop.  Only for the variants that
                        // also carry the 9 vertex-gradient accumulators and the colour path (measured 4-6 % faster with
                        // the re-fetch there, 1-5 % slower for the texel-only and silhouette kernels)
                        const float *rp = A.rec + ((size_t)n * F + f) * REC;
#ifndef UMR_HOST_SHIM
                        asm volatile("" : "+s"(rp));
#endif
                        reload_face(fc, rp);
                    }
                    if (mine < 0) continue;
#if FM_QUADS
                    float xp, yp;
                    unsigned pn4, gp4;
                    if constexpr (QSLOT16) {
                        xp = q4.x + qlxf; yp = q4.y - qlyf;     // exact: small integers over a power of two
#if FM_PACKED
                        pn4 = (unsigned)__float_as_int(q4.z) + qlo_st;
#else
                        pn4 = (unsigned)__float_as_int(q4.z) + qlo_pn;
#endif
                        gp4 = (unsigned)__float_as_int(q4.w);   // (COMMON: the gradient arrives pooled -- the quad IS one pooled pixel)
                    } else {
                        const int xi = (int)(qe & 0xffffu) | qlx, row = (int)(qe >> 16) | qly;
                        if (!(pow2 && IS >= 4) && (xi >= IS || row >= IS)) continue;      // (a power-of-two image has no ragged sub-tile)
                        yp = ndc_coord_fast(IS - 1 - row, IS, inv_is, pow2);
                        xp = ndc_coord_fast(xi, IS, inv_is, pow2);
#if FM_PACKED
                        pn4 = qst + qlo_st;                                           // byte offset of the pixel in the packed state
                        gp4 = pooled ? (unsigned)(UMR_MUL24(row >> 1, H2) + (xi >> 1)) * 4u : (unsigned)(UMR_MUL24(row, IS) + xi) * 4u;
#else
                        pn4 = (unsigned)(UMR_MUL24(row, IS) + xi) * 4u;               // byte offset in a full plane (< 2^24 pixels a side)
                        gp4 = pooled ? (unsigned)(UMR_MUL24(row >> 1, H2) + (xi >> 1)) * 4u : pn4;
#endif
                    }
#else
                    const int row = (mine >> 16) * FM_TH + sl / FM_TW;
                    const int xi = (mine & 0xffff) * FM_TW + sl % FM_TW;
                    if (xi >= IS || row >= IS) continue;
                    const float yp = ndc_coord_fast(IS - 1 - row, IS, inv_is, pow2);
                    const float xp = ndc_coord_fast(xi, IS, inv_is, pow2);
                    const unsigned pn4 = (unsigned)(row * IS + xi) * 4u;                       // byte offset in a full plane
                    const unsigned gp4 = pooled ? (unsigned)((row >> 1) * H2 + (xi >> 1)) * 4u : pn4;
#endif
                    // Exact tile skips from the saved forward state, before any geometry:
                    //  * alpha term: a pixel with alpha == 1.0f exactly contributes g*(1-alpha)*finite = 0 (:584);
                    //  * colour term: p = D*exp((zn - max)/gamma)/S (:608) is 0.0f when even the face's nearest depth
                    //    is >= 89 gamma behind the pixel's soft-max maximum (hard mode: the face is not the winner).
                    // the pixel's saved maximum and alpha: read once, used by the dead test and by the terms (the compiler does not
                    // merge the two reads itself)
                    float smx = 0.f, sal = 0.f;
                    {
                        bool dead;
                        if (RGB == 2) {
                            sal = ld_u(sc_n, pn4);
                            dead = sal == 1.f;
                        } else {
                            dead = false;
                            if (!NEED_GF || AG) {   // (with vertex gradients both terms must vanish: too rare to pay for)
#if FM_PACKED
                                smx = ld_ui<STATE_O_MAX * 4u>(st_n, pn4);
#if FM_DEAD_EAGER
                                sal = ld_ui<STATE_O_ALPHA * 4u>(st_n, pn4);    // both words in flight before the first is tested
#endif
#else
                                smx = ld_u(ag_n, pn4 + pst);
#endif
                                const float zmin_f = fminf(fminf(fc.template g<R_Z0>(), fc.template g<R_Z1>()), fc.template g<R_Z2>());
                                dead = RGB == 0 ? (float)f != smx
                                                : ((c_far - zmin_f) * c_rr - smx) * c_ig < -89.f;
#if FM_PACKED && FM_DEAD_EAGER
                                dead = dead & (sal == 1.f);
#elif FM_PACKED
                                sal = ld_ui<STATE_O_ALPHA * 4u>(st_n, pn4);
                                dead = dead & (sal == 1.f);
#else
                                if (AG) { sal = ld_u(sc_n, pn4 + 3 * pst); dead = dead & (sal == 1.f); }
#endif
                            }
                        }
#if FM_PACKED
                        if (wave_all(dead)) continue;
#else
                        if ((RGB == 2 || !NEED_GF || AG) && __all(dead)) continue;
#endif
                    }
                    Pair p;
                    if (!eval_pair(p, fc, xp, yp, c_thr2, c_nis, A.amb_thr)) continue;
                    if (RGB == 2) {  // silhouette: d alpha only (:584, :632-642); soft_colors/grad are [N,IS,IS] | [N,H,H]
                        if (!fc.depth_in_range()) {
                            float u0, u1, u2;
                            const float zq = clip_depth(u0, u1, u2, p, fc);
                            if (zq < c_near || zq > c_far) continue;  // :592
                        }
                        const float ga = (pooled ? 0.25f : 1.f) * ld_u(gc_n, gp4);
                        UMR_TRAP_IF(umr_bad(ga), 3);
                        const float oa = sal;
                        float c_a = ga * ((1.f - oa) * __builtin_amdgcn_rcpf(fmaxf(1.f - p.frag, 1e-6f)));
                        c_a *= p.frag * (1.f - p.frag) * (-c_nis);
                        const float k2a = 2.f * p.sign * c_a;
                        const float a0 = k2a * p.b0, a1 = k2a * p.b1, a2 = k2a * p.b2;
                        FM_ACC(gv[0], a0, p.dx); FM_ACC(gv[1], a0, p.dy);
                        FM_ACC(gv[3], a1, p.dx); FM_ACC(gv[4], a1, p.dy);
                        FM_ACC(gv[6], a2, p.dx); FM_ACC(gv[7], a2, p.dy);
                        continue;
                    }
                    const float gscale = pooled ? 0.25f : 1.f;   // 2x2 mean pool: each fine pixel gets a quarter
#if FM_PACKED
                    const float g0 = gscale * ld_u(gc_n, gp4), g1 = gscale * ld_u(gc_n, gp4 + gps), g2 = gscale * ld_u(gc_n, gp4 + 2 * gps);
                    const float g3 = gscale * ld_u(gc_n, gp4 + 3 * gps);
#else
                    const float g0 = gscale * ld_u(gc_n, gp4), g1 = gscale * ld_u(gc_n, gp4 + gps),
                                g2 = gscale * ld_u(gc_n, gp4 + 2 * gps);
                    const float g3 = NEED_GF ? gscale * ld_u(gc_n, gp4 + 3 * gps) : 0.f;
#endif
                    UMR_TRAP_IF(umr_bad(g0) | umr_bad(g1) | umr_bad(g2) | umr_bad(g3), 3);
#if FM_PACKED
                    const float rsum = ld_u(st_n, pn4), smax = smx;   // (rsum: the sum's v_rcp_f32, taken by the forward)
                    float c_xy = g3 * ((1.f - sal) * __builtin_amdgcn_rcpf(fmaxf(1.f - p.frag, 1e-6f)));  // :584
#else
                    const float ssum = ld_u(ag_n, pn4), smax = (!NEED_GF || AG) ? smx : ld_u(ag_n, pn4 + pst);
                    const float rsum = __builtin_amdgcn_rcpf(ssum);
                    float c_xy = 0.f;
                    if (NEED_GF) c_xy = g3 * ((1.f - (AG ? sal : ld_u(sc_n, pn4 + 3 * pst))) * __builtin_amdgcn_rcpf(fmaxf(1.f - p.frag, 1e-6f)));  // :584
#endif
                    float q0, q1, q2;
                    const float zp = clip_depth(q0, q1, q2, p, fc);
                    if (zp < c_near || zp > c_far) continue;  // :592
                    float gz0 = 0.f, gz1 = 0.f, gz2 = 0.f;
                    if (RGB == 0) {
                        if (NEED_GT && (float)f == smax) {  // :596
                            const int tix = texel_index(q0, q1, A.R);
                            if (TS == 1) { gt0 += g0; gt1 += g1; gt2 += g2; }
                            else texel_accumulate(my_tex, tix, g0, g1, g2, lane);
                        }
                    } else if (two_sided || fc.front()) {
                        const float zn = div_r(c_far - zp, c_far - c_near, c_rr);
                        // (exponent clamped at 0: zn <= smax for every pair the forward included, so this changes no bit of a legitimate
                        // pair; a pair only the backward's cull kept -- profiles/r04_nan_replay.md -- then weighs at most D / S
                        // instead of exp(9300) = inf)
                        const float ex_ = (zn - smax) * c_ig;                                     // (NaN stays NaN: fminf would swallow it)
                        const float ps = p.frag * __expf(ex_ > 0.f ? 0.f : ex_) * rsum;  // :608
                        const int tix = texel_index(q0, q1, A.R);
                        if (NEED_GT) {
                            if (TS == 1) { FM_ACC(gt0, ps, g0); FM_ACC(gt1, ps, g1); FM_ACC(gt2, ps, g2); }
                            else texel_accumulate(my_tex, tix, ps * g0, ps * g1, ps * g2, lane);
                        }
                        if (NEED_GF && !AG) {
                            const char *tx = (const char *)tex_f;
                            const unsigned t12 = (unsigned)tix * 12u;
                            float c_rgb = g0 * (ld_u(tx, t12) - ld_u(sc_n, pn4));
                            c_rgb += g1 * (ld_u(tx, t12 + 4) - ld_u(sc_n, pn4 + pst));
                            c_rgb += g2 * (ld_u(tx, t12 + 8) - ld_u(sc_n, pn4 + 2 * pst));
                            c_rgb *= ps;
                            c_xy += c_rgb * __builtin_amdgcn_rcpf(p.frag);
                            const float c_z = -(c_rgb * c_ig * c_rr) * zp * zp;  // :624
                            gz0 = c_z * q0 * fc.template g<R_RZ0>() * fc.template g<R_RZ0>();
                            gz1 = c_z * q1 * fc.template g<R_RZ1>() * fc.template g<R_RZ1>();
                            gz2 = c_z * q2 * fc.template g<R_RZ2>() * fc.template g<R_RZ2>();
                        }
                    }
                    if (NEED_GF) {
                        c_xy *= p.frag * (1.f - p.frag) * (-c_nis);  // :632
                        const float k2 = 2.f * p.sign * c_xy;        // :640
                        const float b0 = k2 * p.b0, b1 = k2 * p.b1, b2 = k2 * p.b2;
                        FM_ACC(gv[0], b0, p.dx); FM_ACC(gv[1], b0, p.dy);
                        FM_ACC(gv[3], b1, p.dx); FM_ACC(gv[4], b1, p.dy);
                        FM_ACC(gv[6], b2, p.dx); FM_ACC(gv[7], b2, p.dy);
                        if (!AG) { gv[2] += gz0; gv[5] += gz1; gv[8] += gz2; }
                    }
                }
            }
        }
    }
#ifndef FM_SKIP_EMPTY
#define FM_SKIP_EMPTY 1   // a face none of whose sub-tiles survived the culling pass (half the mesh under a texel-gradient
#endif                    // launch) adds exact zeros: skip its lane reductions, LDS read-out and read-modify-write stores
    // (the item word is fetched again rather than kept across the walk: a wave-uniform value held through the visit loop is an
    // SGPR spill there)
    if (FM_WAVES == 1 && A.order)
        item = (unsigned)__builtin_amdgcn_readfirstlane((int)A.order[((size_t)((blockIdx.x >> 3) / A.order_stride) * 8 + (blockIdx.x & 7)) * A.order_stride +
                                                                     (blockIdx.x >> 3) % A.order_stride].x);
    if (FM_WAVES == 1 && (item >> 21)) {
        // One part of a split face: leave this item's partial sums in its slab (vertex gradients at [0, 9), texel gradients
        // from [16]).  k_split_reduce -- the next launch on the stream -- adds a face's parts in part order
        // and stores: the sum a face gets is a function of its parts' sums and their fixed order alone.  (An in-kernel hand-over
        // to the last arriving item was measured first: its device-scope fences write back / invalidate the XCD's L2 and cost
        // 3x the kernel's time.)
        const int part = (int)(item >> 26);
        const int xcd_ = blockIdx.x & 7, slot_ = blockIdx.x >> 3;
        const unsigned slab0 = (unsigned)__builtin_amdgcn_readfirstlane(
            (int)A.order[((size_t)(slot_ / A.order_stride) * 8 + xcd_) * A.order_stride + slot_ % A.order_stride].y);
        float *sl = A.slab + (size_t)(slab0 + (unsigned)part) * A.slab_stride;
        if (!visited) {        // nothing under this part: its sums are zeros (k_split_reduce adds every part, no flags to fetch first)
            for (int j = lane; j < 16 + (NEED_GT ? TS * 3 : 0); j += 64) sl[j] = 0.f;
            return;
        }
        if (NEED_GF) {
            float mine = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float sv = wave_sum_full(gv[k]);
                if (lane == k) mine = sv;
            }
            if (lane < 9) sl[lane] = mine;
        }
        if (NEED_GT) {
            if (TS == 1) {
                const float s0 = wave_sum_full(gt0), s1 = wave_sum_full(gt1), s2 = wave_sum_full(gt2);
                if (lane < 3) sl[16 + lane] = lane == 0 ? s0 : (lane == 1 ? s1 : s2);
            } else {
                __syncthreads();
                for (int j = lane; j < TS * 3; j += 64) {
                    float acc = wave_tex[j];
#pragma unroll
                    for (int c = 1; c < FM_TEXCOPY; ++c) acc += wave_tex[c * FM_TEX_STRIDE(TS) + j];
                    sl[16 + j] = acc;
                }
            }
        }
        return;
    }
    if (FM_SKIP_EMPTY && FM_WAVES == 1 && !visited) return;
    if (NEED_GF) {
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float sv = wave_sum_full(gv[k]);
            if (lane == k) mine = sv;
        }
        UMR_TRAP_AT(umr_bad(mine), 4 | (RGB == 2 ? 0x40 : (RGB == 0 ? 0x80 : 0)), ((unsigned)(A.N > 32) << 23) | ((unsigned)(n & 127) << 16) | ((unsigned)f & 0xffffu));
        if (live && lane < 9) A.grad_faces[((size_t)n * F + f) * 9 + lane] += mine;
    }
    if (NEED_GT) {
        if (TS == 1) {
            const float s0 = wave_sum_full(gt0), s1 = wave_sum_full(gt1), s2 = wave_sum_full(gt2);
            if (live && lane < 3) A.grad_textures[((size_t)n * F + f) * 3 + lane] += lane == 0 ? s0 : (lane == 1 ? s1 : s2);
        } else {
            __syncthreads();  // every wave arrives exactly once; orders the LDS atomics before the read-out
            if (live) {
                float *dst = A.grad_textures + ((size_t)n * F + f) * TS * 3;
                for (int j = lane; j < TS * 3; j += 64) {
                    float acc = wave_tex[j];
#pragma unroll
                    for (int c = 1; c < FM_TEXCOPY; ++c) acc += wave_tex[c * FM_TEX_STRIDE(TS) + j];
                    UMR_TRAP_AT(umr_bad(acc), 5 | (RGB == 0 ? 0x80 : 0), ((unsigned)(A.N > 32) << 23) | ((unsigned)(n & 127) << 16) | ((unsigned)f & 0xffffu));
                    dst[j] += acc;
                }
            }
        }
    }
}

