// Textual include INSIDE the F(4x4,3x3) kernels of conv_wino4.hip (both main loops share it): the tile epilogue.
// Needs in scope: a, late_args, KArgs, tile_of, acc[18], s_x, s_epi, s_red, cog, ph, lane, wave, H, W and the template
// parameters STATS / ADD / BN, and a constexpr bool W4_SIGNED_TILES and a bool w4_neg (W4_SIGNED_TILES: this block's accumulators hold -M).
    // ---- tile epilogue: output transform Y = A^T M A.  This wave holds M[i = 3 ph + il][j] for (channels cog*16 + kq*4 .. +3,
    // tile l15); A^T over j in registers, then the partial sums over its three rows i for all four output rows; the
    // pair (ph 0, ph 1) of a channel group swaps the two output rows the other one owns (ph 0: rows 0-1, ph 1: rows 2-3).
    float *xw = s_x + (cog * 8 * 64 + lane) * 4;                 // this pair's exchange area: [slot 8][lane 64] float4
    auto tile_epilogue = [&](int k) {
        // wave-uniform coefficients of the half transform, selected HERE (behind an opaque copy of ph): selected once in
        // front of the loop they occupy ten vector registers across it
        int ph_e = ph;
        asm volatile("" : "+s"(ph_e));
        const float cf_s = ph_e == 0 ? 1.f : 4.f, cf_t0 = ph_e == 0 ? 1.f : 0.f;
        const float cf_d = ph_e == 0 ? 1.f : 8.f, cf_t1 = ph_e == 0 ? 0.f : 1.f, cf_h = ph_e == 0 ? 1.f : 2.f;
        if (W4_ABL & 32) {
#pragma unroll
            for (int p = 0; p < 18; ++p) { asm volatile("" ::"v"(acc[p])); acc[p] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            return;
        }
        const int tile_ = tile_of(k);
        const int tx = tile_ % a.tiles_x, ty = (tile_ / a.tiles_x) % a.tiles_y, b = tile_ / (a.tiles_x * a.tiles_y);
        const float tsign = (W4_SIGNED_TILES && w4_neg) ? -1.f : 1.f;
        const KArgs la = late_args();
        const float *e_addend = la->epi.addend, *e_z = la->epi.z, *e_act = la->epi.act;
        float *e_out = la->out;
        float yo[2][4][4], ys[2][4][4];         // partial outputs [output row][column][channel]: kept / handed over
        // A^T restricted to this position half.  ph 0 holds rows (m0, m1, m2), ph 1 holds rows (m5, m3, m4) -- in that
        // order, so that with s = t1 + t2, d = t1 - t2 both halves evaluate the same expressions with wave-uniform
        // coefficients:   ph 0 finishes Y0 = t0 + s, Y1 = d           and hands over Y2 = s, Y3 = d
        //                 ph 1 finishes Y2 = 4 s,    Y3 = 8 d + t0    and hands over Y0 = s, Y1 = 2 d
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float tm[3][4];
#pragma unroll
            for (int il = 0; il < 3; ++il) {
                float m[6], o[4];
#pragma unroll
                for (int j = 0; j < 6; ++j) m[j] = acc[il * 6 + j][r];
                at6(m, o);
#pragma unroll
                for (int j = 0; j < 4; ++j) tm[il][j] = o[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sm = tm[1][j] + tm[2][j], df = tm[1][j] - tm[2][j];
                yo[0][j][r] = fmaf(cf_s, sm, cf_t0 * tm[0][j]);
                yo[1][j][r] = fmaf(cf_d, df, cf_t1 * tm[0][j]);
                ys[0][j][r] = sm;
                ys[1][j][r] = cf_h * df;
            }
        }
        // own output rows 2 ph + io.  Everything behind the output transform runs in the PIXEL-QUAD layout: lane L' =
        // (tile t2 = L' >> 2, channel quad L' & 3), i.e. four ADJACENT lanes own one pixel's 64 contiguous bytes of this
        // wave's 16 channels -- the memory pipeline merges a lane quad's 16-byte pieces into one request, whereas in the
        // accumulator layout (lane = kq*16 + tile) every piece of every epilogue load and store was a request of its own
        // (stores alone: -5 % on the launch).  The transformed outputs move there through four ds_bpermute per pixel.
        // The operands of one output row (4 pixels x up to 3 tensors) are requested together, the first row's before the
        // exchange so that the exchange hides part of their round trip.
        const int t2 = lane >> 2, cq = cog * 16 + (lane & 3) * 4;
        const int oy0 = ty * TH + 4 * (t2 >> 3) + 2 * ph, ox0 = tx * TW + 4 * (t2 & 7);
        const size_t img = (size_t)b * H * W * 64;
        struct Ops { float4 ad[4], z4[4], a4[4]; };
        const bool bits = BN == 2 && la->epi.act_bits != nullptr;       // (wave-uniform)
        auto offs = [&](int io, int j) {        // in-image element offset of the pixel's channels (clamped: loads are unconditional)
            return (unsigned)((min(oy0 + io, H - 1) * W + min(ox0 + j, W - 1)) * 64 + cq);
        };
        auto load_ops = [&](int io, Ops &o) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned off = offs(io, j);
                o.ad[j] = o.z4[j] = o.a4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ADD) o.ad[j] = *reinterpret_cast<const float4 *>(e_addend + img + off);
                if (BN == 1 || BN == 2) o.z4[j] = *reinterpret_cast<const float4 *>(e_z + img + off);
                if (BN == 2) {
                    if (bits) {       // bit (c & 31) of word c >> 5 of the pixel = the decision for channel c
                        const uint32_t wd = la->epi.act_bits[((size_t)b * H * W * 64 + off) >> 5] >> (cq & 31);
                        o.a4[j] = make_float4((float)(wd & 1u), (float)((wd >> 1) & 1u), (float)((wd >> 2) & 1u),
                                              (float)((wd >> 3) & 1u));
                    } else {
                        o.a4[j] = *reinterpret_cast<const float4 *>(e_act + img + off);
                    }
                }
            }
        };
        Ops ops;
        W4_STAMP(16 + 2 * (k - 4), 0);
        if (ADD || BN) load_ops(0, ops);
        // exchange (three barriers; the area is idle between two iterations): ph 1 writes, ph 0 reads and writes into the
        // same slots, ph 1 reads
        float4 rx[2][4];
        auto put = [&]() {
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4 *>(xw + (io * 4 + j) * 256) = make_float4(ys[io][j][0], ys[io][j][1], ys[io][j][2], ys[io][j][3]);
        };
        auto get = [&]() {
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int j = 0; j < 4; ++j) rx[io][j] = *reinterpret_cast<const float4 *>(xw + (io * 4 + j) * 256);
        };
        W4_STAMP(16 + 2 * (k - 4), 1);
        if (ph == 1) put();
        if (!(W4_ABL & 512)) lds_barrier();
        if (ph == 0) { get(); put(); }
        if (!(W4_ABL & 512)) lds_barrier();
        if (ph == 1) get();
        if (!(W4_ABL & 512)) lds_barrier();
        W4_STAMP(16 + 2 * (k - 4), 2);
        // complete the own rows right away (the received values die here): rows are summed in the order i = 0..5 on both
        // waves, (rows 0-2) + (rows 3-5)
#pragma unroll
        for (int io = 0; io < 2; ++io)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float got[4] = {rx[io][j].x, rx[io][j].y, rx[io][j].z, rx[io][j].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) yo[io][j][r] = ph == 0 ? yo[io][j][r] + got[r] : got[r] + yo[io][j][r];
            }
        float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int io = 0; io < 2; ++io) {
            const int oy = oy0 + io;
            float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f}, msc[4] = {0.f, 0.f, 0.f, 0.f}, msh[4] = {0.f, 0.f, 0.f, 0.f};
            if (BN) {           // per-channel constants from LDS, just before their use
                const float4 m4 = *reinterpret_cast<const float4 *>(s_epi + cq), i4 = *reinterpret_cast<const float4 *>(s_epi + 64 + cq);
                mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
                is[0] = i4.x; is[1] = i4.y; is[2] = i4.z; is[3] = i4.w;
                if (BN == 1 || BN == 3) {
                    const float4 s4 = *reinterpret_cast<const float4 *>(s_epi + 128 + cq), h4 = *reinterpret_cast<const float4 *>(s_epi + 192 + cq);
                    msc[0] = s4.x; msc[1] = s4.y; msc[2] = s4.z; msc[3] = s4.w;
                    msh[0] = h4.x; msh[1] = h4.y; msh[2] = h4.z; msh[3] = h4.w;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = oy < H && ox0 + j < W;
                // lane L' takes the transformed outputs of (tile L' >> 2, channel quad L' & 3) from lane (L' & 3) * 16 + (L' >> 2)
                const int src = ((lane & 3) * 16 + (lane >> 2)) * 4;
                float own[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    own[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, yo[io][j][r])));
                const float adv[4] = {ops.ad[j].x, ops.ad[j].y, ops.ad[j].z, ops.ad[j].w};
                const float zv[4] = {ops.z4[j].x, ops.z4[j].y, ops.z4[j].z, ops.z4[j].w};
                const float av[4] = {ops.a4[j].x, ops.a4[j].y, ops.a4[j].z, ops.a4[j].w};
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = W4_SIGNED_TILES ? own[r] * tsign : own[r];      // (split loop: odd tile rows were multiplied with -U)
                    if (BN == 3) {                  // inference: BatchNorm (running statistics) + residual + ReLU
                        v = fmaf(msc[r], v, msh[r]);
                        if (ADD) v += adv[r];
                        if (la->epi.inf_relu) v = v > 0.f ? v : 0.f;
                        o[r] = v;
                        continue;
                    }
                    if (ADD) v += adv[r];
                    if (BN) {
                        const float gate = BN == 2 ? av[r] : fmaf(msc[r], zv[r], msh[r]);
                        if (!(gate > 0.f) || !ok) v = 0.f;
                        ssum[r] += v;
                        ssq[r] += v * ((zv[r] - mu[r]) * is[r]);
                    } else if (STATS) {
                        if (!ok) v = 0.f;
                        ssum[r] += v;
                        ssq[r] += v * v;
                    }
                    o[r] = v;
                }
                if (!(W4_ABL & 256) && ok)
                    *reinterpret_cast<float4 *>(e_out + img + (unsigned)(((oy0 + io) * W + ox0 + j) * 64 + cq)) =
                        make_float4(o[0], o[1], o[2], o[3]);
            }
            if ((ADD || BN) && io == 0) {
                __builtin_amdgcn_sched_barrier(0);
                load_ops(1, ops);
            }
        }
        W4_STAMP(16 + 2 * (k - 4), 3);
        if (STATS) {        // this tile's sums over the wave's 16 tiles (the lanes of one channel quad: L' & 3) -> the wave's running
                            // totals (lanes 0..3)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ssum[r] = quad_class_sum(ssum[r]);
                ssq[r] = quad_class_sum(ssq[r]);
            }
            if (lane < 4) {
                float4 *ps = reinterpret_cast<float4 *>(s_red + wave * 32 + lane * 4), *pq = ps + 4;
                float4 x = *ps, y = *pq;
                x.x += ssum[0]; x.y += ssum[1]; x.z += ssum[2]; x.w += ssum[3];
                y.x += ssq[0]; y.y += ssq[1]; y.z += ssq[2]; y.w += ssq[3];
                *ps = x;
                *pq = y;
            }
        }
        // the accumulators restart from zero -- set here, behind the epilogue, whose operands need the 72 registers
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 18; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
