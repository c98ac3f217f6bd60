// Textual include INSIDE the F(4x4,3x3) kernels of conv_wino4.hip: the per-block statistics row and the BatchNorm tail.
// Needs in scope: late_args, KArgs, s_red, lds (scratch of the fold), tid.
    if (STATS) {        // one partial row [sum 64 | second kind 64] per block; a channel's two position halves are added
        const KArgs la = late_args();
        __syncthreads();
        if (tid < 128) {
            const int kind = tid >> 6, ch = tid & 63, cg = ch >> 4, c = ch & 15;
            bn_tail_store(la->stat_part + (size_t)blockIdx.x * 128 + tid, s_red[cg * 32 + kind * 16 + c] + s_red[(4 + cg) * 32 + kind * 16 + c]);
        }
        BnTail tl;
        tl.mode = la->tail.mode; tl.counter = la->tail.counter; tl.count = la->tail.count;
        tl.gamma = la->tail.gamma; tl.beta = la->tail.beta;
        tl.running_mean = la->tail.running_mean; tl.running_var = la->tail.running_var;
        tl.num_batches_tracked = la->tail.num_batches_tracked; tl.momentum = la->tail.momentum; tl.eps = la->tail.eps;
        tl.scale = la->tail.scale; tl.shift = la->tail.shift; tl.mean = la->tail.mean; tl.invstd = la->tail.invstd;
        tl.dgamma = la->tail.dgamma; tl.dbeta = la->tail.dbeta; tl.abc = la->tail.abc;
        bn_tail_run(tl, la->stat_part, (int)gridDim.x, reinterpret_cast<double *>(lds));
    }
