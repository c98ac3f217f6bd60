// Textual include of conv_wino4.hip (inside its anonymous namespace): the F(4x4,3x3) convolution with the transform-
// domain products on the bf16 matrix pipe -- the "split" main loop (round 5).
//
// Why: on gfx950 the f32 MFMA shares the vector ALU's issue port (tools/probe/mfma_probe.hip), so in the f32 main loop
// the 288 MFMAs of an 8-channel chunk (2,304 cycles per SIMD) and the transform arithmetic ADD UP (4,300 cycles measured).
// v_mfma_f32_16x16x32_bf16 runs at 16x the f32 rate beside the vector ALU.  Each f32 operand is taken as three
// round-to-nearest bf16 pieces x = x0 + x1 + x2 (|x - x0 - x1 - x2| <= 2^-26 |x|) and the six products of order <= 2 are
// accumulated in f32 -- products of bf16 pieces are exact in f32, the three dropped ones are <= 2^-25 |uv|: f32-class
// arithmetic (as conv1 since round 4), 2.67x less matrix time, on a pipe the transforms do not share.
//
// What changes against the f32 loop (same tile, same accumulator layout, same epilogue -- conv_wino4_epi.h):
//   * the MFMA's K is 32 INPUT CHANNELS (a "K-step"; two per tile) instead of 4: the transformed input of a whole K-step
//     would be 36 positions x 16 tiles x 32 channels x 3 pieces x 2 B = 110 KB, so a K-step is cut into three STEPS by
//     transform row: step s carries row i = s of the upper half (wave ph = 0) and row i = 5,3,4 [s] of the lower half
//     (ph = 1) -- 12 positions = 36 KB of pieces, double buffered;
//   * the input transform runs in REGISTERS: thread = (Winograd tile, channel pair, position half) reads its 5 x 6 patch
//     pixels once per K-step (ds_read_b64: two channels), keeps the column stage (3 rows x 6 columns x 2 channels = 36
//     registers) and per step forms one row (bt6), splits the six channel PAIRS into packed bf16 pieces (11 operations a
//     pair, bf3.h) and stores 18 dwords [position][piece][tile][32 channels] -- the MFMA's B operand is then one
//     ds_read_b128 per piece (16-byte slots XOR-swizzled by tile group: conflict-free for the four lane groups of b128);
//   * the transformed weights arrive as their three pieces in the A-operand register image (prep kernel), 54 KB per
//     wave and K-step from L2, three positions ahead of their use;
//   * one LDS-only barrier per step (6 per tile instead of 8), the next K-step's column stage spread over steps 1 and 2.
// LDS: 4 plane slots (two K-steps) of 340 px x 16 channels + 64 B (the two slots of a K-step start 16 banks apart) +
// 2 x 36 KB of pieces = 161,024 B (+ 2.8 KB of per-channel tables: exactly 160 KB).  The pair exchange of the epilogue
// uses piece buffer 1, idle at a tile boundary.
namespace w4s {
constexpr int SLOT_BYTES = w4::NPIX * 64 + 64;            // 21,824
constexpr int PLANES_BYTES = 4 * SLOT_BYTES;              // 87,296
constexpr int VBLK = 1024;                                // one (position, piece): [tile 16][32 channels] bf16
constexpr int VBUF_BYTES = 12 * 3 * VBLK;                 // 36,864
constexpr int LDS_BYTES = PLANES_BYTES + 2 * VBUF_BYTES;  // 161,024
constexpr int U_KSTEP_BYTES = 8 * 18 * 3 * 1024;          // [wave 8][position 18][piece 3][lane 64][8 bf16]
constexpr int U_DWORDS = 2 * U_KSTEP_BYTES / 4;           // 221,184 dwords per convolution and direction
}  // namespace w4s

// Ablation builds (COVA_EXTRA_FLAGS=-DW4S_ABL=<mask>; 0 in the product): 1 no transform (reads, arithmetic, stores),
// 4 no MFMAs, 8 no plane copies, 16 no weight loads, 32 no tile epilogue, 64 no operand reads, 128 weight loads always from
// position 0 (L1 hits), 256 weight operand read from LDS instead of global memory, 512 patch reads without their 2-way bank
// conflict (both tiles of a 32-lane group read one patch; timing only: 0.422 vs 0.420 ms -- the conflict costs nothing)
#ifndef W4S_ABL
#define W4S_ABL 0
#endif
// weight positions in flight ahead of the one being multiplied; the register ring has W4S_LEAD + 1 slots (must divide 36)
#ifndef W4S_LEAD
#define W4S_LEAD 3
#endif

__device__ __forceinline__ f32x4 mfma16bf(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool STATS, int PRO, bool ADD, int BN>
__global__ __launch_bounds__(w4::THREADS, 1) void conv3x3_c64_wino4s_kernel(const W4Args a)
{
    using namespace w4;
    static_assert(PRO != 2, "the two-tensor prologue runs on the f32 main loop");
    typedef const W4Args __attribute__((address_space(4))) *KArgs;
    auto late_args = [&]() -> KArgs {
        unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return (KArgs)kp;
    };
    __shared__ __attribute__((aligned(256))) float lds[w4s::LDS_BYTES / 4];
    __shared__ __attribute__((aligned(16))) float s_epi[BN ? 256 : 4];
    __shared__ __attribute__((aligned(16))) float s_pro[PRO ? 192 : 4];
    __shared__ __attribute__((aligned(16))) float s_red[STATS ? 8 * 32 : 4];
    char *const s_planes = reinterpret_cast<char *>(lds);
    char *const s_vb = s_planes + w4s::PLANES_BYTES;
    float *s_x = reinterpret_cast<float *>(s_vb + w4s::VBUF_BYTES);      // exchange area of the epilogue: piece buffer 1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cog = wave & 3, ph = wave >> 2;
    const int l15 = lane & 15, kq = lane >> 4;
    const int H = a.H, W = a.W;
    if (PRO && tid < 192) s_pro[tid] = late_args()->pro_abc[tid];
    if (BN && tid >= 256) {
        const int c = tid & 63, kind = (tid >> 6) & 3;
        const KArgs la = late_args();
        const float *src = kind == 0 ? la->epi.mean : kind == 1 ? la->epi.invstd : kind == 2 ? la->epi.msc : la->epi.msh;
        const bool used = BN == 3 ? kind >= 2 : (kind < 2 || BN == 1);
        s_epi[kind * 64 + c] = used ? src[c] : 0.f;
    }
    if (PRO || BN) __syncthreads();

    // ---- the tiles of this block.  The block tiles (8 x 32 pixels) are walked as TWO lists -- the tile rows of even ty and those
    // of odd ty -- by the blocks of even and of odd blockIdx.x: a block multiplies with ONE weight image for its whole life (+U on
    // even tile rows, -U on odd ones: see `ubase` below), the sign of a tile depends on its position in the map only (results do
    // not depend on the batch a page sits in), and -- workgroups going to the eight XCDs in turn -- the blocks of an XCD all use
    // the same image: each L2 holds one of the two.  XCD x walks one contiguous quarter of every slice of its list.
    // (w4_grid() on the host keeps gridDim.x even and every block busy; a map of ONE tile row has no odd list.)
    const int Gd = (int)gridDim.x;
    const bool lists = a.tiles_y > 1;
    const int w4_par = lists ? (int)(blockIdx.x & 1) : 0;
    const int Gp = lists ? Gd >> 1 : Gd;
    int bp = lists ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    if ((Gd & 7) == 0) bp = lists ? (int)((blockIdx.x & 7) >> 1) * (Gd >> 3) + (int)(blockIdx.x >> 3)
                                  : (int)(blockIdx.x & 7) * (Gd >> 3) + (int)(blockIdx.x >> 3);
    const int nrow = lists ? (w4_par ? a.tiles_y >> 1 : (a.tiles_y + 1) >> 1) : a.tiles_y;      // tile rows of a page in this list
    const int per_page = nrow * a.tiles_x;
    const int lenp = (a.ntiles / (a.tiles_x * a.tiles_y)) * per_page;
    const int nk = bp < lenp ? (lenp - bp + Gp - 1) / Gp : 0;
    auto tile_of = [&](int k) {
        const int m = bp + (k < nk ? k : nk - 1) * Gp;
        const int b_ = m / per_page, rem = m - b_ * per_page, j = rem / a.tiles_x, tx_ = rem - j * a.tiles_x;
        return (b_ * a.tiles_y + (lists ? 2 * j + w4_par : j)) * a.tiles_x + tx_;
    };
    const bool w4_neg = w4_par != 0;

    // ---- plane copies (global -> LDS, 16 channels = 64 contiguous bytes per pixel and lane quad).  A slot = one 16-channel
    // group of the tile's 340 halo pixels = 22 wave instructions of 16 pixels; wave w issues j = w, w + 8, w + 16 (j < 22),
    // one of them per step.  K-step kk lives in slots 2 (kk & 1), 2 (kk & 1) + 1.
    const unsigned pl_base = (unsigned)(size_t)(lds_void *)s_planes;
    const float *tbase = a.in;
    unsigned rel16[3] = {0, 0, 0};
    bool in16[3] = {false, false, false};
    const int q16 = lane & 3;
    auto px16 = [&](int i) { return 16 * (wave + 8 * i) + (lane >> 2); };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int px = px16(i), r = px / PW, c = px - r * PW;
        rel16[i] = (unsigned)((r * W + c) * 64 + 4 * q16) * 4u;
    }
    auto plane_src = [&](int k) {
        const int tile_ = tile_of(k);
        const int tx = tile_ % a.tiles_x, ty = (tile_ / a.tiles_x) % a.tiles_y, b = tile_ / (a.tiles_x * a.tiles_y);
        const long long org = ((long long)b * H * W + (long long)(ty * TH - 1) * W + (tx * TW - 1)) * 64;
        tbase = a.in + org;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int px = px16(i);
            asm volatile("" : "+v"(px));
            const int r = (px * 241) >> 13, c = px - r * PW;          // px / 34 for px < 352
            const int gy = ty * TH + r - 1, gx = tx * TW + c - 1;
            in16[i] = gy >= 0 && gy < H && gx >= 0 && gx < W;
        }
    };
    auto copy_piece = [&](const int gi, const int slot, const int i) {      // 16-channel group gi of the tile of plane_src()
        if (W4S_ABL & 8) return;
        const int j = wave + 8 * i;
        if (j < 22 && px16(i) < NPIX) {
            const unsigned dst = pl_base + (unsigned)slot * w4s::SLOT_BYTES + j * 1024;
            if (in16[i]) copy16_to_lds(tbase + 16 * gi, rel16[i], dst);
            else copy16_to_lds(g_w4_zero_page, (unsigned)q16 * 16u, dst);
        }
    };

    // ---- transform role: thread = (Winograd tile tt of the 16, channel pair cp of the K-step's 16, position half hf):
    // wave w takes half hf = w & 1 (wave-uniform: the column stage's two forms are scalar branches) of tiles 4 (w >> 1) ..
    // + 3.  Patch rows hf .. hf + 4 (half 0 needs rows 0-4, half 1 rows 1-5), all six columns.
    const int hf = wave & 1;
    const int tt = 4 * (wave >> 1) + (lane >> 4), cp = lane & 15;
    const int tt_rd = (W4S_ABL & 512) ? (tt & ~1) : tt;      // 512: both tiles of a 32-lane group read ONE patch (no bank conflict; timing only)
    const char *rd = s_planes + (cp >> 3) * w4s::SLOT_BYTES + ((4 * (tt_rd >> 3) + hf) * PW + 4 * (tt_rd & 7)) * 64 + (cp & 7) * 8;
    // pieces: block (position of the step 6 half + j, piece q) = [tile 16][64 B]; the 16-byte slot of channels 8 kq .. 8 kq + 7
    // sits at slot kq ^ g(tile >> 2), g = (0, 3, 2, 1): the four 16-lane groups of a ds_read_b128 then cover all 16 slots
    char *vw = s_vb + hf * (6 * 3 * w4s::VBLK) + tt * 64 + ((((cp >> 2) ^ ((4 - (tt >> 2)) & 3))) << 4) + (cp & 3) * 4;
    const char *vr = s_vb + ph * (6 * 3 * w4s::VBLK) + l15 * 64 + ((kq ^ ((4 - (l15 >> 2)) & 3)) << 4);
    // PRO: which of the thread's 5 x 6 patch pixels lie inside the image (bit r * 6 + c), for the tile being staged
    unsigned m30 = 0x3FFFFFFFu;
    bool edge = false;
    auto masks = [&](int k) {
        const int tile_ = tile_of(k);
        const int tx_ = tile_ % a.tiles_x, ty_ = (tile_ / a.tiles_x) % a.tiles_y;
        edge = ty_ == 0 || ty_ * TH + TH + 1 > H || tx_ == 0 || tx_ * TW + TW + 1 > W;       // (block-uniform)
        const int gy0 = ty_ * TH + 4 * (tt >> 3) + hf - 1, gx0 = tx_ * TW + 4 * (tt & 7) - 1;
        unsigned cm = 0, m = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) cm |= (gx0 + c >= 0 && gx0 + c < W) ? (1u << c) : 0u;
#pragma unroll
        for (int r = 0; r < 5; ++r) m |= (gy0 + r >= 0 && gy0 + r < H) ? (cm << (6 * r)) : 0u;
        m30 = m;
    };

    float T[3][6][2];           // column stage: [row of the half][patch column][channel of the pair]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 6; ++c) T[i][c][0] = T[i][c][1] = 0.f;
    // column c of the K-step whose planes sit in slot pair `par`, channels 32 ks + 2 cp, + 1
    auto column = [&](const int c, const int par, const int ks, auto edge_t) {
        constexpr bool EDGE = decltype(edge_t)::value;
        if (W4S_ABL & 1) return;
        const char *p = rd + par * (2 * w4s::SLOT_BYTES) + c * 64;
        float x[5][2];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const float2 v = *reinterpret_cast<const float2 *>(p + r * (PW * 64));
            x[r][0] = v.x; x[r][1] = v.y;
        }
        if (PRO) {
            const float2 A2 = *reinterpret_cast<const float2 *>(s_pro + 32 * ks + 2 * cp);
            const float2 C2 = *reinterpret_cast<const float2 *>(s_pro + 128 + 32 * ks + 2 * cp);
            const float Av[2] = {A2.x, A2.y}, Cv[2] = {C2.x, C2.y};
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float v = fmaf(Av[e], x[r][e], Cv[e]);          // same expression as the f32 loop's prologue
                    if (a.pro_relu) v = fmaxf(v, 0.f);
                    if (EDGE && !((m30 >> (r * 6 + c)) & 1u)) v = 0.f;       // zero padding stays zero
                    x[r][e] = v;
                }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // rows of B^T restricted to the half, on x = d[hf .. hf + 4]; the same expressions as bt6()
            const float o0 = fmaf(4.f, x[0][e], fmaf(-5.f, x[2][e], x[4][e]));       // row 0 (half 0) / row 5 (half 1)
            float o1, o2;
            if (hf == 0) {
                const float u = fmaf(-4.f, x[2][e], x[4][e]), v = fmaf(-4.f, x[1][e], x[3][e]);
                o1 = u + v; o2 = u - v;                                             // rows 1, 2
            } else {
                const float u = x[3][e] - x[1][e], v = 2.f * (x[2][e] - x[0][e]);
                o1 = u + v; o2 = u - v;                                             // rows 3, 4
            }
            T[0][c][e] = o0; T[1][c][e] = o1; T[2][c][e] = o2;
        }
    };
    auto column_e = [&](const int c, const int par, const int ks) {
        if (PRO && edge) column(c, par, ks, std::true_type{});
        else column(c, par, ks, std::false_type{});
    };
    // row sn of the half over the six columns -> positions (sn, j = 0..5) of both channels
    auto row_stage = [&](const int sn, float (&o)[6][2]) {
        if (W4S_ABL & 1) return;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float d[6] = {T[sn][0][e], T[sn][1][e], T[sn][2][e], T[sn][3][e], T[sn][4][e], T[sn][5][e]};
            float r[6];
            bt6(d, r);
#pragma unroll
            for (int j = 0; j < 6; ++j) o[j][e] = r[j];
        }
    };
    auto split_write = [&](const int j, const int parw, const float (&o)[6][2]) {
        if (W4S_ABL & 1) return;
        uint32_t q0, q1, q2;
        bf3_split_pair(o[j][0], o[j][1], q0, q1, q2);
        char *d = vw + parw * w4s::VBUF_BYTES + j * (3 * w4s::VBLK);
        *reinterpret_cast<uint32_t *>(d) = q0;
        *reinterpret_cast<uint32_t *>(d + w4s::VBLK) = q1;
        *reinterpret_cast<uint32_t *>(d + 2 * w4s::VBLK) = q2;
    };

    // ---- weights: position n = 18 ks + p of the 36-position stream is three 1 KB rows (pieces) of this wave's image;
    // wave-uniform base + this lane's fixed byte offset
    // Odd tile rows are multiplied with the image of -U and un-negated in the epilogue: the bf16 MFMA's accumulation is not
    // sign-symmetric (its adder drops low product bits toward -inf: every output came out 7e-8 of the mean magnitude low,
    // whatever its sign -- tools/w4s_bias.py, tools/probe/mfma_round_probe2.hip), a COHERENT offset that reductions over a map
    // (BatchNorm sums, weight gradients) add up; alternating the sign by tile row makes it cancel there.  (Round 5 alternated in
    // a checkerboard, (tx + ty) & 1, with every block taking tiles of both colours: both images streamed through every L2,
    // FETCH_SIZE of the plain forward launch 1.96x its input against 1.27x for the f32 loop's single image.)
    const char *const ubase = reinterpret_cast<const char *>(a.u) + (size_t)wave * (18 * 3 * 1024) +
                              (w4_neg ? (size_t)w4::U_SPLIT_DWORDS * 4 : (size_t)0);
    const unsigned ulane = (unsigned)lane * 16u;
    constexpr int UR = W4S_LEAD + 1;
    static_assert(36 % UR == 0, "ring size must divide the 36-position stream");
    u32x4 Ub[UR][3];
    auto load_u = [&](const int n_, const int slot) {        // n_ >= 36: the position belongs to the NEXT tile of the stream
        if (W4S_ABL & 16) return;
        const int n = n_ % 36;
        const char *ub_t = ubase;
        const int ks = (W4S_ABL & 128) ? 0 : n / 18, p = (W4S_ABL & 128) ? 0 : n - 18 * ks;     // 128: always position 0 (L1 hits)
        if (W4S_ABL & 256) {                                         // 256: the operand from LDS (any 3 KB of the piece buffers)
#pragma unroll
            for (int q = 0; q < 3; ++q) Ub[slot][q] = *reinterpret_cast<const u32x4 *>(vr + ((n % 12) * 3 + q) * w4s::VBLK);
            return;
        }
        const char *bp = ub_t + (size_t)ks * w4s::U_KSTEP_BYTES + p * 3072;
        unsigned ul = ulane;
        asm volatile("" : "+v"(ul));
#pragma unroll
        for (int q = 0; q < 3; ++q) Ub[slot][q] = *reinterpret_cast<const u32x4 *>(bp + q * 1024 + ul);
    };

    // ---- prime: the planes of K-steps 0 and 1, the first three weight positions, K-step 0's column stage and its first row
    plane_src(0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) copy_piece(g, g, i);
#pragma unroll
    for (int n = 0; n < UR; ++n)
#pragma unroll
        for (int q = 0; q < 3; ++q) Ub[n][q] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int n = 0; n < W4S_LEAD; ++n) load_u(n, n);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (PRO) masks(0);
    {
#pragma unroll
        for (int c = 0; c < 6; ++c) column_e(c, 0, 0);
        float o[6][2];
        row_stage(0, o);
#pragma unroll
        for (int j = 0; j < 6; ++j) split_write(j, 0, o);
    }
    if (STATS && tid < 256) s_red[tid] = 0.f;
    __syncthreads();

    f32x4 acc[18];
#pragma unroll
    for (int p = 0; p < 18; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One step = transform row S of K-step KS of tile k: 6 positions x 6 MFMAs from piece buffer (3 KS + S) & 1, while the
    // row of the NEXT step is formed into the other buffer; steps 1 and 2 also run the column stage of the next K-step
    // (three patch columns each: the rows it overwrites were consumed by this step's row stage).
    auto step = [&](const int k, auto ks_t, auto s_t) {
        constexpr int KS = decltype(ks_t)::value, S = decltype(s_t)::value;
        constexpr int TI = 3 * KS + S, PAR = TI & 1, NPAR = (KS + 1) & 1, NKS = NPAR;      // next K-step: slot pair, channel half
        // planes of K-step kk + 2 (tile k + 1, channel half KS) -> the slots K-step kk was staged from
        if (KS == 0 && S == 0) plane_src(k + 1);
        copy_piece(2 * KS, 2 * KS, S);
        copy_piece(2 * KS + 1, 2 * KS + 1, S);
        if (PRO && KS == 1 && S == 1) masks(k + 1);
        const char *vp = vr + PAR * w4s::VBUF_BYTES;
        u32x4 Vr[2][3];
        auto read_v = [&](const int j, const int b) {
            if (W4S_ABL & 64) return;
#pragma unroll
            for (int q = 0; q < 3; ++q) Vr[b][q] = *reinterpret_cast<const u32x4 *>(vp + (j * 3 + q) * w4s::VBLK);
        };
        float o[6][2];
        if (W4S_ABL & 64) {
#pragma unroll
            for (int q = 0; q < 3; ++q) Vr[0][q] = Vr[1][q] = u32x4{0u, 0u, 0u, 0u};
        }
        read_v(0, 0);
        if (S != 2) row_stage(S + 1, o);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            constexpr int n0 = TI * 6;
            const int n = n0 + j, ub = n % UR;
            if (j < 5) read_v(j + 1, (j + 1) & 1);
            if (!(W4S_ABL & 4)) {
                f32x4 c = acc[S * 6 + j];
                const u32x4 *U = Ub[ub], *V = Vr[j & 1];
                // the six products as one dependent chain on the position's accumulator, smallest terms first (a chain on one
                // accumulator issues at the pipe's rate, tools/probe/mfma16_probe.hip; two interleaved half-chains + an add
                // measured 30 % slower)
                c = mfma16bf(U[2], V[0], c);
                c = mfma16bf(U[0], V[2], c);
                c = mfma16bf(U[1], V[1], c);
                c = mfma16bf(U[1], V[0], c);
                c = mfma16bf(U[0], V[1], c);
                c = mfma16bf(U[0], V[0], c);
                acc[S * 6 + j] = c;
            }
            load_u(n + W4S_LEAD, (n + W4S_LEAD) % UR);
            // this position's share of the transform work
            if (S == 0) {
                split_write(j, PAR ^ 1, o);
            } else if (S == 1) {
                split_write(j, PAR ^ 1, o);
                if (j >= 3) column_e(j - 3, NPAR, NKS);
            } else {
                if (j < 3) column_e(3 + j, NPAR, NKS);
                if (j == 2) row_stage(0, o);
                if (j >= 3) { split_write(2 * (j - 3), PAR ^ 1, o); split_write(2 * (j - 3) + 1, PAR ^ 1, o); }
            }
        }
        lds_barrier();
    };

    constexpr bool W4_SIGNED_TILES = true;
#include "conv_wino4_epi.h"

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
    for (int k = 0; k < nk; ++k) {
        step(k, I0{}, I0{});
        step(k, I0{}, I1{});
        step(k, I0{}, I2{});
        step(k, I1{}, I0{});
        step(k, I1{}, I1{});
        step(k, I1{}, I2{});
        tile_epilogue(k);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the copies of the tile past the end have landed before the LDS is released)
#include "conv_wino4_tail.h"
}

// Register image of the transformed weights for the split loop: [K-step ks 2][wave w 8][position p 18][piece q 3][lane 64]
// [e 8] bf16.  Wave w = (cog = w & 3, ph = w >> 2), lane = (kq = lane >> 4, l15 = lane & 15): the MFMA A operand of position
// (i = row il = p / 6 of half ph, j = p % 6) for output channel cog*16 + l15 and input channels 32 ks + 8 kq + e, as the
// three round-to-nearest bf16 pieces of (G g G^T)[i][j].
static_assert(2 * 8 * 18 * 64 * 8 == w4::U_FLOATS, "one thread of prep_wino4_kernel per element of either image");
__device__ void prep_wino4s_element(const float *__restrict__ w, uint16_t *__restrict__ u_fwd, uint16_t *__restrict__ u_dgrad, int idx)
{
    const int e = idx & 7, lane = (idx >> 3) & 63, rest = idx >> 9;
    const int p = rest % 18, wv = (rest / 18) & 7, ks = rest / (18 * 8);
    const int cog = wv & 3, ph = wv >> 2, kq = lane >> 4, l15 = lane & 15;
    const int o = cog * 16 + l15, c = 32 * ks + 8 * kq + e;
    const int il = p / 6, j = p % 6;
    const int i = ph == 0 ? il : (il == 0 ? 5 : 2 + il);
    const float G[6][3] = {{0.25f, 0.f, 0.f},
                           {-1.f / 6, -1.f / 6, -1.f / 6},
                           {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6},
                           {1.f / 24, -1.f / 12, 1.f / 6},
                           {0.f, 0.f, 1.f}};
    double uf = 0.0, ud = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const double coef = (double)G[i][r] * (double)G[j][t];
            uf += coef * (double)w[((o * 64 + c) * 3 + r) * 3 + t];
            ud += coef * (double)w[((c * 64 + o) * 3 + (2 - r)) * 3 + (2 - t)];
        }
    const size_t base = ((((size_t)ks * 8 + wv) * 18 + p) * 3) * 512 + lane * 8 + e;       // piece q: + q * 512
    uint32_t q0, q1, q2;
    bf3_split_pair((float)uf, (float)ud, q0, q1, q2);        // low half: forward, high half: data gradient
    constexpr size_t NEG = 2 * (size_t)w4::U_SPLIT_DWORDS;       // the image of -U (sign bits flipped: the split is sign-symmetric) behind it
    const uint32_t q[3] = {q0, q1, q2};
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3) {
        const uint16_t f = (uint16_t)(q[p3] & 0xFFFFu), d = (uint16_t)(q[p3] >> 16);
        u_fwd[base + 512 * p3] = f;
        u_dgrad[base + 512 * p3] = d;
        u_fwd[NEG + base + 512 * p3] = f ^ 0x8000u;
        u_dgrad[NEG + base + 512 * p3] = d ^ 0x8000u;
    }
}
