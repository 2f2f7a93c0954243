// speller_wide_host.h -- host side of the wide Speller path (speller_wide.h): the per-step launch chains of las_speller_fwd / las_speller_bwd.
// Included by speller.hip behind bwd_layout / make_bf_copies.
#pragma once

#define WIDE_LAUNCH(kernel, grid, block, lds, st, ...)                                                      \
    do {                                                                                                   \
        static int attr__ = wide_lds_attr(kernel, 150 * 1024);                                             \
        LAS_ARG(attr__ == 0 && (size_t)(lds) <= 150 * 1024, "speller (wide): dynamic LDS %zu bytes / attribute %d", (size_t)(lds), attr__); \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                     \
        LAS_LAUNCHED();                                                                                    \
    } while (0)

// which calls the wide path serves: forced by LAS_SPELLER_WIDE wherever the geometry allows; by default the multi-layer and location-aware
// calls that the one-launch loop kernels do not take (until round 5: the per-utterance fp32-operand row kernels) -- in BOTH modes: in parity
// mode it is 62.4 against 85.2 ms (run.sh recipe, rnn cells), 90.0 against 112.3 (lstm), 50.4 against 58.2 (configs[3]); the one-layer
// additive geometry keeps the per-utterance rows there (42.9 against 44.7 ms on the wide path)
template <bool FAST>
static bool wide_selected(const DecDev& d, bool have_ws, bool skinny, bool loop, bool pf) {
    if ((d.flags & LAS_SPELLER_NO_WIDE) || !have_ws || !wide_geom_ok(d)) return false;
    if (FAST && !skinny) return false;
    if (d.flags & LAS_SPELLER_WIDE) return true;
    if (FAST && (d.flags & LAS_SPELLER_NO_BF_ROWS)) return false;         // (the caller asked for the fp32-operand rows in speed mode)
    return !loop && !pf && (d.NL >= 2 || d.mode == LAS_ATT_LOC);
}

template <int CELL, bool FAST>
static int wide_fwd_steps(const las_speller_fwd_args* f, DecDev& d, const BwdWs& wl_, void* packF, hipStream_t st) {
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    const int B = d.B, D = d.D, NL = d.NL, U = d.U, E = d.E, Hd = d.Hd, A = d.A, S = D * NL, GD = G * D, I0D = E + Hd + D;
    const bool loc = d.mode == LAS_ATT_LOC;
    char* wb = (char*)f->ws + wl_.wide;
    const WideWs WL = wide_layout(B, d.Tp, A, D, NL, G, d.C);
    WideDev w;
    wide_fill(d, w, wb, WL);
    if (FAST && !(d.flags & LAS_SPELLER_REUSE_PREP)) {
        GEMM_OK(las_skinny_pack(d.Ws, A, S, A, 0, wb + WL.packWs, st));
        for (int l = 1; l < NL; ++l) GEMM_OK(las_skinny_pack(f->cellW[l], GD, 2 * D, GD, 0, wb + WL.packU[l], st));
    }
    const size_t lds_s = (size_t)(((D + 3) & ~3) + 64) * sizeof(float) + 64;
    const size_t lds_e = wide_lds_bytes(d, w.fper);
    const size_t lds_c = (size_t)(((d.Tp + 3) & ~3) + 4 * RNT) * sizeof(float) + 64;
    // tanh cells, speed mode, tokens known in advance (no in-loop logits): the cell IS the product's epilogue (las_skinny_gemm_bf16_tanh:
    // h = tanh(. + bias) -> the saved state, and as bf16 straight into the operand rows of the products that read it next), so the state
    // launch (after step 0) and the gate launches between the layers' products drop out of the chain: 7 -> 5 dependent launches per step
    const bool epi = FAST && CELL == LAS_CELL_RNN && !d.step_logits && NL <= 2 && !(d.flags & LAS_SPELLER_NO_FUSED_STEP);    // (the flag: every phase its own launch)
    // LSTM cells under the same conditions: product + gate math of a layer in ONE launch (las_skinny_lstm_bf16: the four gate column tiles of 16 units
    // per workgroup), bf16 h straight into the operand rows -- 7 -> 4 dependent launches per forward step
    const bool lepi = FAST && CELL == LAS_CELL_LSTM && !d.step_logits && NL <= 2 && (D % 16) == 0 && !(d.flags & LAS_SPELLER_NO_FUSED_STEP);
    // energies + alignment / context as one launch with an in-kernel hand-over (wide_attend_kernel): every workgroup of the grid resident at once
    const int SP = w.nsplit > w.hsplit ? w.nsplit : w.hsplit;
    const bool fuse = !(d.flags & LAS_SPELLER_NO_FUSED_STEP) && (long long)SP * B <= las_device_cus();
    const size_t lds_f = lds_e > lds_c ? lds_e : lds_c;
    if (fuse) LAS_HIP(hipMemsetAsync(w.egran, 0, (size_t)B * wide_gran_row(d.Tp) * 8, st));
    // an utterance's slices on one XCD (hand-overs through that XCD's L2) when the device deals workgroup ids to its XCDs round-robin and the
    // padded 1-D grid still fits; LAS_SPELLER_NO_PF_ROWS -- a development switch this path has no other use for -- keeps the 2-D grid
    w.sp = SP;
    w.xcd_local = (fuse && las_xcd_round_robin() && (long long)SP * ((B + 7) / 8) * 8 <= las_device_cus() && !(d.flags & LAS_SPELLER_NO_PF_ROWS)) ? 1 : 0;
    const dim3 grid_f = w.xcd_local ? dim3(SP * ((B + 7) / 8) * 8) : dim3(SP, B);
    for (int t = 0; t <= U; ++t) {
        if (!(epi || lepi) || t == 0) WIDE_LAUNCH((wide_state_kernel<CELL, FAST>), dim3(B), dim3(RNT), lds_s, st, d, w, t);
        if (t == U) break;
        if (FAST) GEMM_OK(las_skinny_gemm_bf16(w.sbf, S, B, S, wb + WL.packWs, A, w.qbuf, A, nullptr, st));
        else GEMM_OK(las_gemm(LAS_PREC_F32, 0, 0, B, A, S, 1.f, reinterpret_cast<const float*>(w.sbf), S, 0, d.Ws, A, 0, 0.f, w.qbuf, A, 0, nullptr, LAS_ACT_NONE, 1, 0, 0, nullptr, 0, st));
        if (fuse) {
            if (loc && d.C == 10) WIDE_LAUNCH((wide_attend_kernel<FAST, true, 10>), grid_f, dim3(RNT), lds_f, st, d, w, t);
            else if (loc) WIDE_LAUNCH((wide_attend_kernel<FAST, true>), grid_f, dim3(RNT), lds_f, st, d, w, t);
            else          WIDE_LAUNCH((wide_attend_kernel<FAST, false>), grid_f, dim3(RNT), lds_f, st, d, w, t);
        } else {
            if (loc && d.C == 10) WIDE_LAUNCH((wide_energy_kernel<FAST, true, 10>), dim3(w.nsplit, B), dim3(RNT), lds_e, st, d, w, t);
            else if (loc) WIDE_LAUNCH((wide_energy_kernel<FAST, true>), dim3(w.nsplit, B), dim3(RNT), lds_e, st, d, w, t);
            else     WIDE_LAUNCH((wide_energy_kernel<FAST, false>), dim3(w.nsplit, B), dim3(RNT), lds_e, st, d, w, t);
            WIDE_LAUNCH((wide_context_kernel<FAST>), dim3(w.hsplit, B), dim3(RNT), lds_c, st, d, w, t);
        }
        float* g0 = d.gates + ((size_t)0 * U + t) * B * GD;
        if (lepi) {      // (operand rows as for the tanh epilogue below)
            unsigned short* xu_cur = w.xu + (size_t)(t & 1) * B * 2 * D;
            unsigned short* xu_nxt = w.xu + (size_t)((t + 1) & 1) * B * 2 * D;
            GEMM_OK(las_skinny_lstm_bf16(d.xbf, I0D, B, I0D, packF, D, f->cellb[0], d.fb, d.cs + ((size_t)0 * (U + 1) + t) * B * D,
                                         d.cs + ((size_t)0 * (U + 1) + t + 1) * B * D, d.hs + ((size_t)0 * (U + 1) + t + 1) * B * D, g0,
                                         NL > 1 ? xu_cur : nullptr, 2 * D, w.sbf, S, st));
            if (NL == 2)
                GEMM_OK(las_skinny_lstm_bf16(xu_cur, 2 * D, B, 2 * D, wb + WL.packU[1], D, f->cellb[1], d.fb, d.cs + ((size_t)1 * (U + 1) + t) * B * D,
                                             d.cs + ((size_t)1 * (U + 1) + t + 1) * B * D, d.hs + ((size_t)1 * (U + 1) + t + 1) * B * D,
                                             d.gates + ((size_t)1 * U + t) * B * GD, xu_nxt + D, 2 * D, w.sbf + D, S, st));
            continue;
        }
        if (epi) {
            // layer 0: h_{0,t+1} -> hs[0][t+1]; bf16 into the layer above's [x ; h] row (multi-layer) and into the state row of step t+1
            // (two operand-row buffers for layer 1, alternating per step: its product reads [h_{0,t+1} ; h_{1,t}] from buffer t % 2 while its
            //  epilogue leaves h_{1,t+1} in the upper half of buffer (t + 1) % 2 -- in one buffer a column tile would overwrite what another
            //  tile's K walk still reads)
            unsigned short* xu_cur = w.xu + (size_t)(t & 1) * B * 2 * D;
            unsigned short* xu_nxt = w.xu + (size_t)((t + 1) & 1) * B * 2 * D;
            GEMM_OK(las_skinny_gemm_bf16_tanh(d.xbf, I0D, B, I0D, packF, GD, f->cellb[0], d.hs + ((size_t)0 * (U + 1) + t + 1) * B * D, D,
                                              NL > 1 ? xu_cur : nullptr, 2 * D, w.sbf, S, st));
            if (NL == 2)
                GEMM_OK(las_skinny_gemm_bf16_tanh(xu_cur, 2 * D, B, 2 * D, wb + WL.packU[1], GD, f->cellb[1], d.hs + ((size_t)1 * (U + 1) + t + 1) * B * D, D,
                                                  xu_nxt + D, 2 * D, w.sbf + D, S, st));
            continue;
        }
        if (FAST) {
            GEMM_OK(las_skinny_gemm_bf16(d.xbf, I0D, B, I0D, packF, GD, g0, GD, f->cellb[0], st));
        } else {
            const bool hw = f->ws && f->ws_bytes > wl_.gemm;
            GEMM_OK(las_gemm(f->prec, 0, 0, B, GD, I0D, 1.f, d.xin0 + (size_t)t * B * I0D, I0D, 0, f->cellW[0], GD, 0, 0.f, g0, GD, 0, f->cellb[0],
                             LAS_ACT_NONE, 1, 0, 0, hw ? (char*)f->ws + wl_.gemm : nullptr, hw ? f->ws_bytes - wl_.gemm : 0, st));
        }
        for (int l = 1; l < NL; ++l) {
            hipLaunchKernelGGL((wide_pointwise_fwd_kernel<CELL, FAST>), dim3(B), dim3(256), 0, st, d, w, l - 1, t);
            LAS_LAUNCHED();
            float* gl = d.gates + ((size_t)l * U + t) * B * GD;
            if (FAST) {
                GEMM_OK(las_skinny_gemm_bf16(w.xu, 2 * D, B, 2 * D, wb + WL.packU[l], GD, gl, GD, f->cellb[l], st));
            } else {
                GEMM_OK(las_gemm(f->prec, 0, 0, B, GD, D, 1.f, d.hs + ((size_t)(l - 1) * (U + 1) + t + 1) * B * D, D, 0,
                                 f->cellW[l], GD, 0, 0.f, gl, GD, 0, f->cellb[l], LAS_ACT_NONE, 1, 0, 0, nullptr, 0, st));
                GEMM_OK(las_gemm(f->prec, 0, 0, B, GD, D, 1.f, d.hs + ((size_t)l * (U + 1) + t) * B * D, D, 0,
                                 f->cellW[l] + (size_t)D * GD, GD, 0, 1.f, gl, GD, 0, nullptr, LAS_ACT_NONE, 1, 0, 0, nullptr, 0, st));
            }
        }
    }
    return 0;
}

// the reverse loop (part 1 of las_speller_bwd): leaves dXin0, d(pre-activation) over the gates, dQ, dE, d f, duRows, and dKeys
template <int CELL, bool FAST>
static int wide_bwd_steps(const las_speller_bwd_args* bk, DecDev& d, const BwdWs& wl_, char* base, void* packB, float* dHl, float* tmp,
                          void* gws, size_t gws_bytes, hipStream_t st) {
    constexpr int G = CELL == LAS_CELL_LSTM ? 4 : 1;
    const las_speller_fwd_args* f = &bk->f;
    const int B = d.B, D = d.D, NL = d.NL, U = d.U, E = d.E, Hd = d.Hd, A = d.A, Tp = d.Tp, S = D * NL, GD = G * D, I0D = E + Hd + D, TOP = NL - 1;
    const bool loc = d.mode == LAS_ATT_LOC;
    const int prec = f->prec;
    char* wb = base + wl_.wide;
    const WideWs WL = wide_layout(B, Tp, A, D, NL, G, d.C);
    WideDev w;
    wide_fill(d, w, wb, WL);
    if (FAST) {
        GEMM_OK(las_skinny_pack(d.Ws, A, A, S, 1, wb + WL.packWsT, st));                      // B[k = attention column][n = state row] = Ws[n][k]
        for (int l = 1; l < NL; ++l) GEMM_OK(las_skinny_pack(f->cellW[l], GD, GD, 2 * D, 1, wb + WL.packUB[l], st));
    }
    const size_t lds_a = (size_t)(((Hd + 3) & ~3) + 64) * sizeof(float) + 64;
    const size_t lds_e = wide_lds_bytes(d, w.fper);
    const size_t lds_q = wide_dq_lds_bytes(d, w.fper);
    // speed mode: a layer's gate gradient is the EPILOGUE of the product in front of it (las_skinny_gemm_bf16_tanh_bwd / _lstm_bwd) -- the top
    // layer's behind d s = dq . Ws^T, a lower layer's behind the product of the layer above -- so the gate launches drop out of the chain
    // (8 -> 6 dependent launches per step at two layers; the first iteration, which has no d s, keeps the top layer's launch)
    const bool bepi = FAST && NL <= 2 && !(d.flags & LAS_SPELLER_NO_FUSED_STEP);
    // (one call for both cells: the tanh cell reads h_{l,t+1}, the LSTM cell its activated gates, c_{l,t+1}, c_{l,t} and the carried d c)
    auto gate_epilogue = [&](const unsigned short* Aop, int lda, int K, const void* pack, int N, float* Cout, int ldc, int c0, int layer, int t,
                             const float* sa, int lda_, const float* sb, int ldb, int vlast) -> int {
        float* gp = d.gates + ((size_t)layer * U + t) * B * GD;
        unsigned short* gb = layer == 0 ? d.dgbf : w.dgu;
        if (CELL == LAS_CELL_LSTM)
            return las_skinny_gemm_bf16_lstm_bwd(Aop, lda, B, K, pack, N, Cout, ldc, c0, D, sa, lda_, sb, ldb, vlast, gp, GD, gb, GD,
                                                 d.cs + ((size_t)layer * (U + 1) + t + 1) * B * D, d.cs + ((size_t)layer * (U + 1) + t) * B * D,
                                                 d.dC + (size_t)layer * B * D, D, st);
        return las_skinny_gemm_bf16_tanh_bwd(Aop, lda, B, K, pack, N, Cout, ldc, c0, D, d.hs + ((size_t)layer * (U + 1) + t + 1) * B * D, D,
                                             sa, lda_, sb, ldb, vlast, gp, GD, gb, GD, st);
    };
    // d alpha -> d energy -> dq as ONE launch with two in-kernel hand-overs (wide_attend_bwd_kernel), under the conditions of the forward one
    const bool fuse = !(d.flags & LAS_SPELLER_NO_FUSED_STEP) && (long long)w.nsplit * B <= las_device_cus();
    const size_t lds_f = lds_a > lds_e ? (lds_a > lds_q ? lds_a : lds_q) : (lds_e > lds_q ? lds_e : lds_q);
    if (fuse) LAS_HIP(hipMemsetAsync(w.bgran, 0, (size_t)B * wide_bgran_row(Tp) * 8, st));
    w.sp = w.nsplit;
    w.xcd_local = (fuse && las_xcd_round_robin() && (long long)w.nsplit * ((B + 7) / 8) * 8 <= las_device_cus() && !(d.flags & LAS_SPELLER_NO_PF_ROWS)) ? 1 : 0;
    const dim3 grid_f = w.xcd_local ? dim3(w.nsplit * ((B + 7) / 8) * 8) : dim3(w.nsplit, B);
    for (int t = U - 1; t >= -1; --t) {
        const int ta = t + 1;
        if (ta < U) {    // attention backward of step t + 1 (its context gradient is in dXin0[t + 1])
            if (fuse) {
                if (loc && d.C == 10) WIDE_LAUNCH((wide_attend_bwd_kernel<FAST, true, 10>), grid_f, dim3(RNT), lds_f, st, d, w, ta);
                else if (loc) WIDE_LAUNCH((wide_attend_bwd_kernel<FAST, true>), grid_f, dim3(RNT), lds_f, st, d, w, ta);
                else          WIDE_LAUNCH((wide_attend_bwd_kernel<FAST, false>), grid_f, dim3(RNT), lds_f, st, d, w, ta);
            } else if (loc) {
                WIDE_LAUNCH((wide_dalpha_kernel<FAST, true>), dim3(w.nsplit, B), dim3(RNT), lds_a, st, d, w, ta);
                if (d.C == 10) WIDE_LAUNCH((wide_energy_bwd_kernel<FAST, true, 10>), dim3(w.nsplit, B), dim3(RNT), lds_e, st, d, w, ta);
                else           WIDE_LAUNCH((wide_energy_bwd_kernel<FAST, true>), dim3(w.nsplit, B), dim3(RNT), lds_e, st, d, w, ta);
                WIDE_LAUNCH((wide_dq_kernel<FAST, true>), dim3(w.nsplit, B), dim3(RNT), lds_q, st, d, w, ta);
            } else {
                WIDE_LAUNCH((wide_dalpha_kernel<FAST, false>), dim3(w.nsplit, B), dim3(RNT), lds_a, st, d, w, ta);
                WIDE_LAUNCH((wide_energy_bwd_kernel<FAST, false>), dim3(w.nsplit, B), dim3(RNT), lds_e, st, d, w, ta);
                WIDE_LAUNCH((wide_dq_kernel<FAST, false>), dim3(1, B), dim3(RNT), lds_q, st, d, w, ta);
            }
            if (ta > 0) {    // d s_{t+1} = dq . Ws^T: the gradient of every layer's state that entered step t + 1
                if (bepi && t >= 0) {
                    const float* recT = TOP == 0 ? d.dXin0 + (size_t)ta * B * I0D + E + Hd : tmp + (size_t)TOP * B * 2 * D + D;
                    GEMM_OK(gate_epilogue(w.dqbf, A, A, wb + WL.packWsT, S, w.dS, S, TOP * D, TOP, t, dHl + (size_t)t * B * D, D,
                                          recT, TOP == 0 ? I0D : 2 * D, 1));
                } else if (FAST) GEMM_OK(las_skinny_gemm_bf16(w.dqbf, A, B, A, wb + WL.packWsT, S, w.dS, S, nullptr, st));
                else GEMM_OK(las_gemm(prec, 0, 1, B, S, A, 1.f, d.dQ + (size_t)ta * B * A, A, 0, d.Ws, A, 0, 0.f, w.dS, S, 0, nullptr, LAS_ACT_NONE, 1,
                                      0, 0, nullptr, 0, st));
            }
        }
        if (t < 0) break;
        const bool next = ta < U;
        for (int l = TOP; l >= 0; --l) {
            const float* rec = nullptr; int rec_ld = 0, rec_off = 0;
            if (next) {
                if (l == 0) { rec = d.dXin0 + (size_t)ta * B * I0D; rec_ld = I0D; rec_off = E + Hd; }
                else        { rec = tmp + (size_t)l * B * 2 * D; rec_ld = 2 * D; rec_off = D; }
            }
            const float* extra = l == TOP ? dHl + (size_t)t * B * D : tmp + (size_t)(l + 1) * B * 2 * D;
            const int extra_ld = l == TOP ? D : 2 * D;
            unsigned short* gb = FAST ? (l == 0 ? d.dgbf : w.dgu) : nullptr;
            if (!(bepi && (l < TOP || next))) {              // (else: done by the epilogue of the product in front)
                hipLaunchKernelGGL((wide_cell_bwd_kernel<CELL, FAST>), dim3(B), dim3(256), 0, st, d, w, l, t, rec, rec_ld, rec_off,
                                   next ? (const float*)w.dS : (const float*)nullptr, extra, extra_ld, gb);
                LAS_LAUNCHED();
            }
            const float* dG = d.gates + ((size_t)l * U + t) * B * GD;
            if (l == 0) {
                if (FAST) GEMM_OK(las_skinny_gemm_bf16(d.dgbf, GD, B, GD, packB, I0D, d.dXin0 + (size_t)t * B * I0D, I0D, nullptr, st));
                else GEMM_OK(las_gemm(prec, 0, 1, B, I0D, GD, 1.f, dG, GD, 0, f->cellW[0], GD, 0, 0.f, d.dXin0 + (size_t)t * B * I0D, I0D, 0, nullptr,
                                      LAS_ACT_NONE, 1, 0, 0, gws, gws_bytes, st));
            } else {
                float* tl = tmp + (size_t)l * B * 2 * D;
                if (bepi) {      // + the gate gradient of layer l - 1 (its d h: columns [0, D) of this product, + recurrent + attention shares)
                    const int lb = l - 1;
                    const float* recL = !next ? nullptr : (lb == 0 ? d.dXin0 + (size_t)ta * B * I0D + E + Hd : tmp + (size_t)lb * B * 2 * D + D);
                    GEMM_OK(gate_epilogue(w.dgu, GD, GD, wb + WL.packUB[l], 2 * D, tl, 2 * D, 0, lb, t, recL, lb == 0 ? I0D : 2 * D,
                                          next ? (const float*)w.dS + (size_t)lb * D : (const float*)nullptr, S, 0));
                } else if (FAST) GEMM_OK(las_skinny_gemm_bf16(w.dgu, GD, B, GD, wb + WL.packUB[l], 2 * D, tl, 2 * D, nullptr, st));
                else GEMM_OK(las_gemm(prec, 0, 1, B, 2 * D, GD, 1.f, dG, GD, 0, f->cellW[l], GD, 0, 0.f, tl, 2 * D, 0, nullptr, LAS_ACT_NONE, 1, 0, 0,
                                      nullptr, 0, st));
            }
        }
    }
    {   // keys gradient (and the Wf-gradient partials) contracted over the steps
        const size_t lds_k = loc ? (size_t)(((d.C * A + 3) & ~3) + 8 * d.C * A) * sizeof(float) : 0;
        if (loc && d.C == 10) WIDE_LAUNCH((wide_dkeys_kernel<FAST, true, 10>), dim3(cdiv(Tp, 8), B), dim3(256), lds_k, st, d, w, bk->d_keys);
        else if (loc) WIDE_LAUNCH((wide_dkeys_kernel<FAST, true>), dim3(cdiv(Tp, 8), B), dim3(256), lds_k, st, d, w, bk->d_keys);
        else     WIDE_LAUNCH((wide_dkeys_kernel<FAST, false>), dim3(cdiv(Tp, 8), B), dim3(256), lds_k, st, d, w, bk->d_keys);
    }
    return 0;
}
