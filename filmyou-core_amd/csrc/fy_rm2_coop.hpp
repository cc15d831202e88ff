// fy_rm2_coop.hpp -- host orchestration of the cooperative multi-rank path (fy_collectives, include/filmyou.h).
// NOT a standalone header: one section of fy_rm2.hip's translation unit (needs fy_rm2_job, ScoreTune, Plan and the kernels),
// split out for size.
#pragma once

// ================================================================ cooperative ranks: one cluster scored by all ranks together
// (include/filmyou.h, fy_collectives).  score(u, i) = pvpi + sum over the user's rated items j of a term that
// needs only row j of M, so the sum splits over any partition of the item rows: rank r builds rows r, r + world, ... of M (and of
// the block maxima), evaluates for EVERY user of the cluster the partial sums over the rated items that fall into its
// rows, and a reduce-scatter hands the owner of each user the complete sums.  Three exchanges follow the three pruned
// passes: seed columns, block bounds, surviving blocks.  Rank 0 contributes pvpi; a rated candidate is masked (NaN) by
// the rank that holds its row, and NaN survives the sum.
struct CoopShared {
    fy_rm2_job* J;
    fy_result* R;
    const ScoreTune* tune;
    const float *b_rank;              // b_i (fp32), rank order
    const float *a_rank, *csc_x, *csr_x, *csr_e, *csr_q;
    const uint32_t* csr_pk;        // packed CSR for the row kernel (nullptr: csr_idx / csr_x); csc_x then holds x / s_v
    const int32_t *n_out, *out_off;   // this rank's users, by slot - lo
    const double* pvpi;               // this rank's users, by slot - lo (the complete value; pv_all below is zero on ranks != 0)
    int32_t lo;
    EventTimer *t_cooc, *t_score, *t_topn;
    unsigned long long* prune_counters;
    int64_t *blocks_total, *seed_terms_cols, *coop_survived, *coop_pair_contribs;
    const int32_t* cshift;            // [cluster]: c of the packed matrix format (k_user_meta)
    float gscale;                     // 2^-c of THIS cluster
};

static void coll_all_gather(fy_rm2_job* J, const void* send, void* recv, int64_t bytes, hipStream_t st) {
    if (!J->have_coll) {   // world == 1 (forced cooperative mode, tests): the identity
        if (bytes) FY_HIP(hipMemcpyAsync(recv, send, (size_t)bytes, hipMemcpyDeviceToDevice, st));
        return;
    }
    const int rc = J->coll.all_gather(J->coll.user, send, recv, bytes, (void*)st);
    if (rc) FY_FAIL(FY_ERR_COLLECTIVE, "all_gather callback returned %d", rc);
}
static void coll_reduce_scatter(fy_rm2_job* J, const float* send, float* recv, int64_t count, hipStream_t st) {
    if (!J->have_coll) {
        if (count) FY_HIP(hipMemcpyAsync(recv, send, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, st));
        return;
    }
    const int rc = J->coll.reduce_scatter_f32(J->coll.user, send, recv, count, (void*)st);
    if (rc) FY_FAIL(FY_ERR_COLLECTIVE, "reduce_scatter callback returned %d", rc);
}

static void score_cluster_coop(const CoopShared& X, const Plan& p, hipStream_t ls) {
    fy_rm2_job* J = X.J;
    Context* ctx = J->ctx;
    Prepared& P = J->P;
    fy_result* R = X.R;
    const ScoreTune& tune = *X.tune;
    const fy_rm2_params& prm = J->prm;
    const int W = prm.world, me = prm.rank;
    const int32_t Uc = p.Uc, sbase = p.sbase, pbase = p.pbase, Ic = p.Ic, CH = p.CH, nch = p.nch;
    const int64_t ldm = p.ldm, ldb = p.ldb;
    const double lambda = prm.lambda;

    // ---- who owns which users of this cluster (identical on every rank)
    std::vector<int32_t> ua(W), ub(W);
    int32_t Umax = 1;
    for (int k = 0; k < W; k++) {
        int32_t lo_k, hi_k;
        owner_range(J, k, lo_k, hi_k);
        ua[k] = std::max(lo_k, sbase);
        ub[k] = std::max(ua[k], std::min(hi_k, sbase + Uc));
        Umax = std::max(Umax, ub[k] - ua[k]);
    }
    const int32_t my_a = ua[me], n_mine = ub[me] - ua[me];

    // ---- item rows of this rank: me, me + W, me + 2 W, ... (popularity order: every rank gets the same mix of rows)
    const int32_t r0 = me, nrows = Ic > me ? (Ic - me + W - 1) / W : 0;
    std::vector<int32_t> hcnt((size_t)Ic);
    int64_t my_ratings = 0, my_walk = 0;
    {
        std::vector<long long> hw((size_t)Ic);
        FY_HIP(hipMemcpyAsync(hw.data(), J->walk_rank.get() + pbase, (size_t)Ic * sizeof(long long), hipMemcpyDeviceToHost, ls));
        FY_HIP(hipMemcpyAsync(hcnt.data(), J->cnt_rank.get() + pbase, (size_t)Ic * sizeof(int32_t), hipMemcpyDeviceToHost, ls));
        FY_HIP(hipStreamSynchronize(ls));
        for (int32_t i = r0; i < Ic; i += W) { my_ratings += hcnt[i]; my_walk += hw[i]; }
    }
    *X.coop_pair_contribs += my_walk - my_ratings;   // ordered off-diagonal co-rating pairs whose row is mine

    // ---- buffers
    const int n_chunks = (int)ceil_div(Ic, 256);
    const int seed_chunks = std::min(n_chunks, tune.seed_chunks);
    const int seed_blocks = seed_chunks;
    const int64_t SC = (int64_t)seed_chunks * 256;
    const int bchunks = (int)(ldb / 256);
    DevBuf<float> Mloc(ctx, (size_t)std::max<int64_t>(1, (int64_t)nrows * ldm * 3 / 4 + 4)), Bloc(ctx, (size_t)std::max<int64_t>(1, (int64_t)nrows * ldb));
    DevBuf<float> amax(ctx, (size_t)ldb), bmax(ctx, (size_t)ldb);
    DevBuf<int32_t> n_out_all(ctx, (size_t)Uc), item_counter(ctx, 1);
    // my compact CSR over all users of the cluster (local row indices); its size is not known on the host: room for all
    const int64_t nnz_c = (int64_t)p.nq;
    DevBuf<int32_t> my_cnt(ctx, (size_t)Uc + 1), my_rowptr(ctx, (size_t)Uc + 1), my_idx(ctx, (size_t)std::max<int64_t>(1, nnz_c));
    DevBuf<float> my_e(ctx, (size_t)std::max<int64_t>(1, nnz_c)), my_q(ctx, (size_t)std::max<int64_t>(1, nnz_c));
    DevBuf<double> pv_all(ctx, (size_t)Uc);
    DevBuf<unsigned long long> dummy(ctx, 2);
    DevBuf<float> seed_send(ctx, (size_t)((int64_t)W * Umax * SC)), seed(ctx, (size_t)((int64_t)Umax * SC));
    DevBuf<float> ub_send(ctx, (size_t)((int64_t)W * Umax * ldb)), UBsum(ctx, (size_t)((int64_t)Umax * ldb));
    DevBuf<float> tau(ctx, (size_t)Umax);
    DevBuf<uint16_t> surv(ctx, (size_t)((int64_t)Umax * ldb));
    DevBuf<int32_t> n_quads(ctx, (size_t)Umax + 1), quad_prefix(ctx, (size_t)Umax + 1), overflow(ctx, (size_t)Umax), any_overflow(ctx, 1);
    DevBuf<int32_t> counts(ctx, (size_t)W);
    const float* Mshift = Mloc.get();    // M / Bmax are indexed by the LOCAL row (k-th row of this rank)
    const float* Bshift = Bloc.get();

    // ---- segment table of my rows (over a compact copy of their CSC entries), M build
    {
        DevBuf<int32_t> co_tmp(ctx, (size_t)Uc * (nch + 1));
        SegTable seg;
        DevBuf<int2> item_seg(ctx, (size_t)std::max<int64_t>(1, (int64_t)nrows * nch));
        DevBuf<int32_t> item_id(ctx, (size_t)std::max<int64_t>(1, (int64_t)nrows * nch));
        build_chunk_offsets(ctx, P.rowptr.get(), P.csr_idx.get(), sbase, Uc, CH, nch, co_tmp.get(), ls);
        std::vector<int32_t> hls((size_t)nrows + 1, 0);
        for (int32_t i = 0; i < nrows; i++) hls[i + 1] = hls[i] + hcnt[r0 + (int64_t)i * W];
        DevBuf<int32_t> local_start(ctx, (size_t)nrows + 1), my_slot(ctx, (size_t)std::max<int64_t>(1, my_ratings));
        DevBuf<float> my_w(ctx, (size_t)std::max<int64_t>(1, my_ratings));
        FY_HIP(hipMemcpyAsync(local_start.get(), hls.data(), ((size_t)nrows + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ls));
        if (nrows > 0) {
            k_gather_rows<<<grid_for((int64_t)nrows * 64, 256), 256, 0, ls>>>(r0, W, nrows, P.rank_pair.get() + pbase, P.pair_start.get(),
                                                                              local_start.get(), P.csc_slot.get(), X.csc_x, my_slot.get(), my_w.get());
            FY_KERNEL_CHECK();
        }
        build_segments(ctx, my_slot.get(), my_w.get(), co_tmp.get(), sbase, 0, (int32_t)my_ratings, nch, seg, ls);
        FY_HIP(hipMemsetAsync(Bloc.get(), 0, (size_t)std::max<int64_t>(1, (int64_t)nrows * ldb) * 3, ls));
        k_block_amax<<<grid_for(ldb), 256, 0, ls>>>(Ic, (int32_t)ldb, X.a_rank + pbase, X.b_rank + pbase, amax.get(), bmax.get());
        FY_KERNEL_CHECK();
        if (nrows > 0) {
            CoocArgs CA{P.rank_pair.get(), P.pair_start.get(), seg.ptr.get(), seg.seg.get(), seg.w.get(), P.csr_idx.get(),
                        X.csr_x, pbase, sbase, Ic, CH, nch, r0, nrows, 0, (int32_t)my_ratings, local_start.get(), W, X.csr_pk, nullptr,
                        (uint32_t)std::min<int64_t>((int64_t)P.nnz * 4, 0xFFFFFFFFll)};
            const int fxk = (X.csr_pk && tune.cooc_fx && !J->fx_bounds.empty()) ? fx_exponent(&J->fx_bounds[3 * (size_t)p.c]) : -1;
            CA.fx_scale = fxk >= 0 ? std::ldexp(1.0, fxk) : 0.0;
            const double w2s = (1.0 - lambda) * (1.0 - lambda) * (double)X.gscale;      // (1-l)^2 and the packed format's 2^-c
            MEpilogue ME{const_cast<float*>(Mshift), ldm, (float)w2s, fxk >= 0 ? std::ldexp(w2s, -fxk) : 0.0,
                         1, const_cast<float*>(Bshift), ldb, 1};
            const size_t sp = X.t_cooc->begin(ls);
            const int n_items = nrows * nch;
            k_item_list<<<grid_for(n_items), 256, 0, ls>>>(CA, item_seg.get(), item_id.get());
            FY_KERNEL_CHECK();
            CA.item_seg = item_seg.get();
            CA.item_id = item_id.get();
            FY_HIP(hipMemsetAsync(item_counter.get(), 0, sizeof(int32_t), ls));
            launch_cooc_rm2(ctx, tune, X.csr_pk != nullptr, CA, ME, n_items, item_counter.get(), ls);
            X.t_cooc->end(sp, ls);
            R->st.cooc_launches++;
        }
        FY_HIP(hipStreamSynchronize(ls));   // the segment table is released here
    }

    // ---- per-user tables over the whole cluster
    k_my_csr_count<<<grid_for(((int64_t)Uc + 1) * 64, 256), 256, 0, ls>>>(Uc, sbase, P.rowptr.get(), P.csr_idx.get(), W, me, my_cnt.get());
    FY_KERNEL_CHECK();
    exclusive_scan_i32(ctx, my_cnt.get(), my_rowptr.get(), (size_t)Uc + 1, ls);
    k_my_csr_fill<<<grid_for((int64_t)Uc * 64, 256), 256, 0, ls>>>(Uc, sbase, P.rowptr.get(), P.csr_idx.get(), X.csr_e, X.csr_q, W, me, my_rowptr.get(),
                                                                   my_idx.get(), my_e.get(), my_q.get());
    FY_KERNEL_CHECK();
    FY_HIP(hipMemsetAsync(dummy.get(), 0, 2 * sizeof(unsigned long long), ls));
    k_user_meta<<<grid_for(Uc), 256, 0, ls>>>(sbase, sbase + Uc, P.slot2du.get(), P.uid.get(), P.ucluster.get(), P.udeg.get(),
                                               P.d_csize.get(), P.d_pcstart.get(), prm.number_of_items, prm.number_of_recommendations,
                                               prm.filter_users, X.cshift, pv_all.get(), n_out_all.get(), dummy.get());
    FY_KERNEL_CHECK();
    if (me != 0) FY_HIP(hipMemsetAsync(pv_all.get(), 0, (size_t)Uc * sizeof(double), ls));   // pvpi enters the sum once

    size_t ss = X.t_score->begin(ls);   // the spans of ms_score cover this rank's kernels, not the waits inside the collectives
    auto slices_for = [&](int32_t nb) { return score_slices(ctx, tune, nb, seed_chunks + bchunks); };
    // ---- (1) partial seed scores AND partial block bounds of every user in one launch per owner (one grid tail, see
    // rm2_score); (2) two reduce-scatters; (3) tau + the speculative lists; (5) the owner keeps the blocks that reach tau
    for (int k = 0; k < W; k++) {
        const int32_t nk = ub[k] - ua[k];
        if (nk <= 0) continue;
        const int ns = slices_for(nk);
        ScoreArgs SA{};
        SA.M = Mshift; SA.ldm = ldm; SA.Ic = Ic; SA.a_rank = X.a_rank + pbase; SA.b_rank = X.b_rank + pbase; SA.rb_off = my_rowptr.get();
        SA.csr_idx = my_idx.get(); SA.csr_e = my_e.get(); SA.csr_q = my_q.get(); SA.pvpi = pv_all.get(); SA.n_out = n_out_all.get(); SA.slot_lo = sbase; SA.slot_base = sbase; SA.slot0 = ua[k];
        SA.n_users = nk; SA.S = seed_send.get() + (int64_t)k * Umax * SC; SA.ldS = SC; SA.n_slices = ns; SA.n_chunks = seed_chunks + bchunks;
        SA.row_mul = W; SA.row_add = me;
        SA.chunks1 = seed_chunks;
        SA.M2 = Bshift;
        SA.ldm2 = ldb;
        SA.Ic2 = p.nblk;
        SA.a2 = amax.get();
        SA.b2 = bmax.get();
        SA.S2 = ub_send.get() + (int64_t)k * Umax * ldb;
        SA.ldS2 = ldb;
        SA.no_mask2 = 1;
        k_score<4, true, 8><<<(seed_chunks + bchunks) * ns, 256, 0, ls>>>(SA.M, SA.a_rank, SA.rb_off, SA.csr_idx, SA.csr_e, SA.csr_q, SA.pvpi, SA.n_out, SA.S, SA);
        FY_KERNEL_CHECK();
        R->st.score_launches += 2;
    }
    X.t_score->end(ss, ls);
    coll_reduce_scatter(J, seed_send.get(), seed.get(), (int64_t)Umax * SC, ls);
    coll_reduce_scatter(J, ub_send.get(), UBsum.get(), (int64_t)Umax * ldb, ls);
    ss = X.t_score->begin(ls);
    if (n_mine > 0) {
        TopNArgs T1{seed.get(), SC, Ic, X.n_out, X.out_off, P.rank_item_raw.get() + pbase, P.slot2du.get(), P.uid.get(),
                    X.lo, my_a, p.c, R->d_key0.get(), R->d_key1.get(), R->d_value.get(), R->d_aux.get(),
                    1, (int32_t)SC, surv.get(), n_quads.get(), ldb, tau.get(), nullptr, nullptr};
        k_topn_fast<<<n_mine, 256, 0, ls>>>(T1, overflow.get(), any_overflow.get(), 0);
        FY_KERNEL_CHECK();
    }
    FY_HIP(hipMemsetAsync(n_quads.get(), 0, ((size_t)Umax + 1) * sizeof(int32_t), ls));
    if (n_mine > 0) {
        k_bound_select<<<grid_for((int64_t)n_mine * 64, 256), 256, 0, ls>>>(UBsum.get(), ldb, p.nblk, seed_blocks, tau.get(), X.pvpi + (my_a - X.lo), n_mine,
                                                                            surv.get(), n_quads.get());
        FY_KERNEL_CHECK();
    }
    exclusive_scan_i32(ctx, n_quads.get(), quad_prefix.get(), (size_t)n_mine + 1, ls);
    // ---- (6) everybody learns everybody's survivors
    DevBuf<int32_t> my_count(ctx, 1);
    FY_HIP(hipMemcpyAsync(my_count.get(), quad_prefix.get() + n_mine, sizeof(int32_t), hipMemcpyDeviceToDevice, ls));
    X.t_score->end(ss, ls);
    coll_all_gather(J, my_count.get(), counts.get(), sizeof(int32_t), ls);
    ss = X.t_score->begin(ls);
    std::vector<int32_t> hcounts((size_t)W);
    FY_HIP(hipMemcpyAsync(hcounts.data(), counts.get(), (size_t)W * sizeof(int32_t), hipMemcpyDeviceToHost, ls));
    FY_HIP(hipStreamSynchronize(ls));
    int32_t t_max = 0;
    for (int k = 0; k < W; k++) t_max = std::max(t_max, hcounts[k]);
    DevBuf<float> Ssurv;
    if (t_max > 0) {
        DevBuf<long long> entries(ctx, (size_t)t_max), entries_all(ctx, (size_t)W * t_max);
        DevBuf<float> Spart(ctx, (size_t)W * t_max * PRUNE_BLOCK);
        Ssurv.alloc(ctx, (size_t)t_max * PRUNE_BLOCK);
        FY_HIP(hipMemsetAsync(entries.get(), 0, (size_t)t_max * sizeof(long long), ls));
        if (n_mine > 0) {
            k_surv_entries<<<grid_for((int64_t)n_mine * 64, 256), 256, 0, ls>>>(n_mine, my_a, quad_prefix.get(), surv.get(), ldb, entries.get());
            FY_KERNEL_CHECK();
        }
        X.t_score->end(ss, ls);
        coll_all_gather(J, entries.get(), entries_all.get(), (int64_t)t_max * (int64_t)sizeof(long long), ls);
        ss = X.t_score->begin(ls);
        // ---- (7) partial exact scores of all survivors over my rows, reduce-scatter to the owners
        k_score_entries<8><<<ctx->num_cus * 8, 256, 0, ls>>>(Mshift, X.a_rank + pbase, X.b_rank + pbase, my_rowptr.get(), my_idx.get(), my_e.get(), my_q.get(), pv_all.get(),
                                                             entries_all.get(), counts.get(), W, t_max, sbase, Ic, ldm, W, me, Spart.get(),
                                                             X.prune_counters);
        FY_KERNEL_CHECK();
        R->st.score_launches++;
        X.t_score->end(ss, ls);
        coll_reduce_scatter(J, Spart.get(), Ssurv.get(), (int64_t)t_max * PRUNE_BLOCK, ls);
        ss = X.t_score->begin(ls);
        FY_HIP(hipStreamSynchronize(ls));   // entries / Spart are released here
    }
    X.t_score->end(ss, ls);
    // ---- (8) the owner merges seed + survivors for the users that have any
    if (n_mine > 0 && hcounts[me] > 0) {
        const size_t tt = X.t_topn->begin(ls);
        TopNArgs TA{seed.get(), SC, Ic, X.n_out, X.out_off, P.rank_item_raw.get() + pbase, P.slot2du.get(), P.uid.get(),
                    X.lo, my_a, p.c, R->d_key0.get(), R->d_key1.get(), R->d_value.get(), R->d_aux.get(),
                    2, (int32_t)SC, surv.get(), n_quads.get(), ldb, tau.get(), Ssurv.get(), quad_prefix.get()};
        FY_HIP(hipMemsetAsync(any_overflow.get(), 0, sizeof(int32_t), ls));
        k_topn_fast<<<n_mine, 256, 0, ls>>>(TA, overflow.get(), any_overflow.get(), tune.force_select);
        FY_KERNEL_CHECK();
        k_topn_select<<<n_mine, 256, 0, ls>>>(TA, overflow.get(), any_overflow.get(), X.prune_counters + 2);
        FY_KERNEL_CHECK();
        X.t_topn->end(tt, ls);
    }
    // statistics: blocks checked / kept for my users; log terms of my seed and bound passes = (ratings in my rows) x (columns walked)
    *X.blocks_total += (int64_t)n_mine * std::max(0, p.nblk - seed_blocks);
    *X.coop_survived += (int64_t)hcounts[me];
    *X.seed_terms_cols += my_ratings * (SC + ldb);
    FY_HIP(hipStreamSynchronize(ls));   // every buffer of this function goes back to the allocator after the stream drained
}

