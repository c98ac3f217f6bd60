// Textual include of conv.hip (inside its anonymous namespace, behind conv1_7x7_bf3_kernel whose geometry, LDS images and
// helpers it shares): conv1 forward on the bf16 matrix pipe with ONE wave per SIMD.
//
// conv1_7x7_bf3_kernel runs two waves per SIMD with the same instruction mix (six MFMAs, then a chunk of other work, ...):
// its phase traces showed the older wave of a SIMD through its K loop in 9,400 cycles and the younger 3,000 later -- 8,450
// cycles of matrix-pipe work take 12,900 per tile.  Here a block is FOUR waves: wave q owns output rows 2q, 2q + 1 of the
// 8 x 32 tile for BOTH channel halves (four 32 x 32 accumulators), issues the tile's 264 MFMAs as one stream and puts every
// other instruction of the tile -- the operand reads one K-step ahead, the previous tile's 64 output stores, the next tile's
// patch prefetch, its split into bf16 pieces and its LDS writes -- into the shadow of those MFMAs, one small chunk behind
// each (a bf16 MFMA of 32 cycles leaves room for six vector instructions of the same wave).
//
// MEASURED, NOT THE DEFAULT (cova_set_option(12, 1)): 0.94-0.97 ms against 0.76-0.79 ms of the 8-wave kernel, alternated inside
// the train step (tools/ab_step.py 12), with hand-placed chunks and with sched_group_barrier groups alike -- 59 cycles per MFMA
// instead of 32: the ~6 other instructions per MFMA of this tile (LDS operand reads, stores with their statistics, the
// 30-instruction address computation of a patch slot, accumulator <-> vector register moves: the kernel needs 256 + 243
// registers) do not hide behind the wave's own MFMAs the way a SECOND wave's instructions do.  Same accumulation order as the
// 8-wave kernel: bit-identical output (tests/test_kernels_gpu.py::test_conv1_bf16_split_error_class).
namespace c1w {
constexpr int THREADS = 256;
constexpr int NPRE = (c1b::PLANE + THREADS - 1) / THREADS;       // 9 pair slots per thread
constexpr int RF0 = 6;                                           // K-step at which the prefetched patch starts going to LDS
}  // namespace c1w

template <bool STATS>
__global__ __launch_bounds__(c1w::THREADS) void conv1_7x7_bf3w_kernel(
    const float *__restrict__ img, const float *__restrict__ wk, float *__restrict__ out,
    float *__restrict__ stat_part, int H, int W, int H1, int W1, int tiles_x, int tiles_y, int ntiles,
    int w_oihw, const BnTail tail)
{
    using namespace c1b;
    constexpr int THREADS = c1w::THREADS, NPRE = c1w::NPRE, RF0 = c1w::RF0;
    static_assert(NPRE == 9 && 2 * (KSTEPS - RF0) >= NPRE + 1 && 2 * 5 >= NPRE, "prefetch / refill schedule of the K loop");
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * BUF + 4 * WB_VEC + 4 * 128];
    uint32_t *s_pp = lds;                                        // [buffer][piece][segment][pair]
    u32x4 *s_wb = reinterpret_cast<u32x4 *>(lds + 2 * BUF);      // B operands [K-step][channel block][piece][lane]
    float *s_red = reinterpret_cast<float *>(lds + 2 * BUF + 4 * WB_VEC);     // [wave][sum 64 | sum of squares 64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = row pair
    const int li = lane & 31, kh2 = lane >> 5;
    int tile = blockIdx.x;

    // ---- the next tile's patch: slot it of thread tid = pair (it * 256 + tid) of the [segment][36] grid (as the 8-wave kernel)
    float pre[2 * NPRE];
    unsigned premask = 0u;
    unsigned po0 = 0u, po1 = 0u, pm = 0u;
    auto slot_addr = [&](int it, int ty, int tx) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));          // the slot's index math stays here (not hoisted out of the tile loop)
        const int item = it * THREADS + t_;
        const int itc = item < PLANE ? item : PLANE - 1;
        const int seg = itc / SEGW, m = itc - seg * SEGW;
        const int c = seg / PR, r = seg - c * PR;
        const int gy = 2 * ty * TH - 3 + r, gx = 2 * tx * TW - 3 + 2 * m;
        const bool oky = item < PLANE && gy >= 0 && gy < H;
        const bool ok0 = oky && gx >= 0 && gx < W, ok1 = oky && gx + 1 >= 0 && gx + 1 < W;
        const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
        const int cx0 = gx < 0 ? 0 : (gx >= W ? W - 1 : gx), cx1 = gx + 1 < 0 ? 0 : (gx + 1 >= W ? W - 1 : gx + 1);
        const unsigned rowo = (unsigned)((c * H + cy) * W);              // 32-bit in-image offset (checked at launch)
        po0 = rowo + cx0;
        po1 = rowo + cx1;
        pm = (ok0 ? 1u : 0u) | (ok1 ? 2u : 0u);
    };
    auto slot_load = [&](int it, const float *img_b) {
        pre[2 * it] = img_b[po0];
        pre[2 * it + 1] = img_b[po1];
        premask = (premask & ~(3u << (2 * it))) | (pm << (2 * it));
    };
    uint32_t rq0 = 0u, rq1 = 0u, rq2 = 0u;
    auto refill_split = [&](int it) {
        const float xe = ((premask >> (2 * it)) & 1u) ? pre[2 * it] : 0.f;
        const float xo = ((premask >> (2 * it + 1)) & 1u) ? pre[2 * it + 1] : 0.f;
        bf3_split_pair(xe, xo, rq0, rq1, rq2);
    };
    auto refill_write = [&](int it, uint32_t *dst) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const int item = it * THREADS + t_;
        if (item < PLANE) {
            dst[item] = rq0;
            dst[PLANE + item] = rq1;
            dst[2 * PLANE + item] = rq2;
        }
    };
    if (tile < ntiles) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
#pragma unroll
        for (int it = 0; it < NPRE; ++it) { slot_addr(it, ty, tx); slot_load(it, img + (size_t)b * 3 * H * W); }
    }
    // ---- B-operand image (as the 8-wave kernel): entry (K-step s, channel block, lane (n, g))
    for (int e = tid; e < KSTEPS * 2 * 64; e += THREADS) {
        const int l = e & 63, cbe = (e >> 6) & 1, s = e >> 7;
        const int co = cbe * 32 + (l & 31), t = 2 * s + (l >> 5);
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (t < 21 && j < 7) {
                const int tap = t * 7 + j;                       // c*49 + kh*7 + kw
                if (w_oihw) {
                    v = wk[co * 147 + tap];
                } else {                                         // inverse of c1::pair_tap
                    int row;
                    if (tap < 98) row = 2 * (tap % 49) + tap / 49;
                    else {
                        const int kh = (tap - 98) / 7, kw = (tap - 98) % 7;
                        row = kh < 3 ? 2 * (49 + kh * 7 + kw) : kh == 3 ? 2 * (70 + kw) : 2 * (49 + (kh - 4) * 7 + kw) + 1;
                    }
                    v = wk[row * 64 + co];
                }
            }
            wv[j] = v;
        }
        u32x4 q0, q1, q2;
        bf3_split8(wv, q0, q1, q2);
        s_wb[((s * 2 + cbe) * 3 + 0) * 64 + l] = q0;
        s_wb[((s * 2 + cbe) * 3 + 1) * 64 + l] = q1;
        s_wb[((s * 2 + cbe) * 3 + 2) * 64 + l] = q2;
    }
    if (STATS) { s_red[tid] = 0.f; s_red[tid + 256] = 0.f; }
    if (tile < ntiles) {
#pragma unroll
        for (int it = 0; it < NPRE; ++it) { refill_split(it); refill_write(it, s_pp); }
    }
    __syncthreads();

    // ---- pending tile: accumulators pd[cb][row], stored during the next tile's K loop (64 stores, six per K-step)
    f32x16 pd[2][2];
    bool pend = false;
    float *pb00 = out, *pb01 = out, *pb10 = out, *pb11 = out;            // [row][pixels 0-15 | 16-31] bases of the pending tile (cb 0)
    const unsigned st_off = (unsigned)((4 * kh2) * 64 + li);            // lane: pixel 4*(l >> 5), channel l & 31
    double tot_s[2] = {0.0, 0.0}, tot_q[2] = {0.0, 0.0};                 // per channel block
    float sm[2] = {0.f, 0.f}, sq[2] = {0.f, 0.f};
    auto pend_store = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 64) {
            constexpr int cb = k >> 5, rr = (k >> 4) & 1, r = k & 15;
            constexpr int imm = ((r & 3) + 8 * ((r >> 2) & 1)) * 64 + cb * 32;
            const float v = pd[cb][rr][r];
            (rr == 0 ? ((r >> 3) ? pb01 : pb00) : ((r >> 3) ? pb11 : pb10))[st_off + imm] = v;
            sm[cb] += v;
            sq[cb] = fmaf(v, v, sq[cb]);
        }
    };
    const u32x4 *b_base = s_wb + lane;

    int cur = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x;
        const int ty = (tile / tiles_x) % tiles_y;
        const int b = tile / (tiles_x * tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        const bool has_next = tile + (int)gridDim.x < ntiles;
        const int next = has_next ? tile + (int)gridDim.x : tile;       // (last tile: re-reads its own patch, unused)
        const int ntx = next % tiles_x, nty = (next / tiles_x) % tiles_y;
        const float *nimg = img + (size_t)(next / (tiles_x * tiles_y)) * 3 * H * W;
        const uint32_t *a_org = s_pp + cur * BUF + (4 * q) * SEGW + li;  // output row 2q (+ 2*SEGW: row 2q+1), dwords li..li+3

        f32x16 acc[2][2];                                                // [cb][row]
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int rw = 0; rw < 2; ++rw)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][rw][r] = 0.f;
        // operands double-buffered in registers, fetched one K-step ahead
        u32x4 g1[2][2], g2a[2][2], g2b[2][2], bl[2][2][3];               // [K-step parity][row] / [K-step parity][cb][piece]
        auto a_ptr = [&](int s_) { return a_org + (kh2 ? row_off(2 * s_ + 1) : row_off(2 * s_)); };
        auto load4 = [&](u32x4 &d, const uint32_t *p_) {
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = p_[i];
        };
        auto load_b = [&](int s_, int cb, int pc) { bl[s_ & 1][cb][pc] = b_base[((s_ * 2 + cb) * 3 + pc) * 64]; };
        load4(g1[0][0], a_ptr(0)); load4(g1[0][1], a_ptr(0) + 2 * SEGW);
        load4(g2a[0][0], a_ptr(0) + PLANE); load4(g2a[0][1], a_ptr(0) + 2 * SEGW + PLANE);
        load4(g2b[0][0], a_ptr(0) + 2 * PLANE); load4(g2b[0][1], a_ptr(0) + 2 * SEGW + 2 * PLANE);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) load_b(0, cb, pc);
        uint32_t *rdst = s_pp + (cur ^ 1) * BUF;
        // two instantiations -- with and without a pending tile's stores (straight-line code either way: the compiler's waits
        // for a prefetched value count exactly the loads and stores issued after it)
        auto kloop = [&](auto stc) {
        constexpr bool st = decltype(stc)::value;
        c1b_static_for<KSTEPS>([&](auto sc) {
            constexpr int s = decltype(sc)::value, P = s & 1, N = P ^ 1;
            constexpr bool more = s + 1 < KSTEPS;
            // ---- everything else of this K-step, in program order in front of its MFMAs; the scheduler interleaves it behind
            // them in groups (one MFMA, then up to four vector, one LDS, one memory, one scalar instruction -- the group
            // barriers below): a bf16 MFMA of 32 cycles covers about six issue slots of the same wave
            if constexpr (more) {
                load4(g1[N][0], a_ptr(s + 1));
                load4(g1[N][1], a_ptr(s + 1) + 2 * SEGW);
                load4(g2a[N][0], a_ptr(s + 1) + PLANE);
                load4(g2a[N][1], a_ptr(s + 1) + 2 * SEGW + PLANE);
                load4(g2b[N][0], a_ptr(s + 1) + 2 * PLANE);
                load4(g2b[N][1], a_ptr(s + 1) + 2 * SEGW + 2 * PLANE);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) load_b(s + 1, cb, pc);
            }
            if constexpr (st) {                                           // the pending tile's stores: six per K-step
                c1b_static_for<6>([&](auto jc) { pend_store(std::integral_constant<int, 6 * s + decltype(jc)::value>{}); });
            }
            if constexpr (2 * s < NPRE) { slot_addr(2 * s, nty, ntx); slot_load(2 * s, nimg); }            // the next tile's patch:
            if constexpr (2 * s + 1 < NPRE) { slot_addr(2 * s + 1, nty, ntx); slot_load(2 * s + 1, nimg); }  // requested at K-steps 0..4 ...
            if constexpr (s >= RF0 && 2 * (s - RF0) < NPRE) { refill_split(2 * (s - RF0)); refill_write(2 * (s - RF0), rdst); }
            if constexpr (s >= RF0 && 2 * (s - RF0) + 1 < NPRE) { refill_split(2 * (s - RF0) + 1); refill_write(2 * (s - RF0) + 1, rdst); }
            // 24 MFMAs: products a0 x b2, a0 x b1, a0 x b0, a2 x b0, a1 x b1, a1 x b0 (the 8-wave kernel's order per accumulator);
            // within a product (row, cb) = (0,0) (1,0) (0,1) (1,1): four independent accumulators in turn
            c1b_static_for<24>([&](auto mc) {
                constexpr int m = decltype(mc)::value, prod = m >> 2, row = m & 1, cb = (m >> 1) & 1;
                const u32x4 av = prod < 3 ? g1[P][row] : (prod == 3 ? g2b[P][row] : g2a[P][row]);
                constexpr int bp = prod == 0 ? 2 : prod == 1 ? 1 : prod == 2 ? 0 : prod == 3 ? 0 : prod == 4 ? 1 : 0;
                acc[cb][row] = mfma32bf(av, bl[P][cb][bp], acc[cb][row]);
            });
            c1b_static_for<24>([&](auto) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);       // VALU
                __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);       // LDS
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);       // global memory
                __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);       // SALU
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        };
        if (pend) kloop(std::true_type{}); else kloop(std::false_type{});
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            tot_s[cb] += (double)sm[cb];
            tot_q[cb] += (double)sq[cb];
            sm[cb] = 0.f; sq[cb] = 0.f;
        }
        pend = false;
        c1b_lds_barrier();               // the next patch is complete; every wave is done reading this one
        cur ^= 1;
        // output rows y0 + 2q, + 1: D register r of lane l = pixel mfma32_row(r, l), channel cb*32 + (l & 31)
        if (y0 + TH <= H1 && x0 + TW <= W1) {                    // interior: stored during the next tile (or after the loop)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) { pd[cb][0] = acc[cb][0]; pd[cb][1] = acc[cb][1]; }
            pend = true;
            pb00 = out + (((size_t)b * H1 + y0 + 2 * q) * W1 + x0) * 64;
            pb01 = pb00 + 16 * 64;
            pb10 = pb00 + (size_t)W1 * 64;
            pb11 = pb10 + 16 * 64;
        } else {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int oy = y0 + 2 * q + rr;
                    const size_t rowb = ((size_t)b * H1 + oy) * W1 + x0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int px = mfma32_row(r, lane);
                        if (oy < H1 && x0 + px < W1) {
                            const float v = acc[cb][rr][r];
                            out[(rowb + (size_t)px) * 64 + cb * 32 + li] = v;
                            sm[cb] += v;
                            sq[cb] = fmaf(v, v, sq[cb]);
                        }
                    }
                }
        }
    }
    if (pend) {                          // the block's last tile
        c1b_static_for<64>(pend_store);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        tot_s[cb] += (double)sm[cb];
        tot_q[cb] += (double)sq[cb];
    }
    if (STATS) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            tot_s[cb] += __shfl_xor(tot_s[cb], 32, 64);
            tot_q[cb] += __shfl_xor(tot_q[cb], 32, 64);
            if (lane < 32) {
                s_red[q * 128 + cb * 32 + li] = (float)tot_s[cb];
                s_red[q * 128 + 64 + cb * 32 + li] = (float)tot_q[cb];
            }
        }
    }
    __syncthreads();
    if (STATS && tid < 128) {
        float t = 0.f;
        for (int w = 0; w < 4; ++w) t += s_red[w * 128 + tid];
        bn_tail_store(stat_part + (size_t)blockIdx.x * 128 + tid, t);
    }
    if (STATS) bn_tail_run(tail, stat_part, (int)gridDim.x, reinterpret_cast<double *>(lds));
}
