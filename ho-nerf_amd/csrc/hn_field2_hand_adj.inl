// Body of k_field2_hand<2> after the forward pass of a sample tile: the adjoint of the evaluation
// (oracle/field_bwd.py steps 3b-6), included inside the kernel's tile loop so that it shares the forward pass's
// helpers (feature_pass, load_bone, block_epilogue, coords ...).  What it reads from the wave's stash -- the tape --
// was written by the forward pass in this mode: a1..a8, dz0..dz7, c1..c4, the per-bone sums of d sdf / d features,
// the feature fragments.
//
//   colour network backward      cb_l = relu'(c_l) * (C_{l+1}^T cb_{l+1});  C0^T cb1 -> fb | gb | X-adjoint (per bone)
//   forward-direction sweep      GXb = J gb;  dzb_l = W_l (sigma'_{l-1} dzb_{l-1}) [+ W4x GXb];  w_l = sigma''_l u_l dzb_l
//   second reverse sweep         zb_l = sigma'_l ab_l + w_l;  ab_{l-1} = W_l^T zb_l;  X-adjoint += W0^T zb0 + W4x^T zb4
//   input map, bone by bone      qbar_b = d/dq (X-adjoint . features) + Hessian-vector term;  g_pts += R_b^T qbar_b;
//                                g_bt_inv[b] += qbar_b (x) [p, 1] + d/dq(GX) (x) gb;  g_T_pose[b] -= qbar_b
//
// Every adjoint quantity is linear in the upstream gradients; they are scaled per sample by a power of two kappa
// (largest seed in [1, 2)) so that the fp16 hi/lo fragments keep their 22 bits at any loss scale; 1 / kappa (exact)
// comes back at the outputs.
{
    struct Act2 {
        f32x16 v;   // stashed activation a_{l+1}: sigma'(z_l) = 1 - exp(-100 a)
        f32x16 x;   // dz_l (forward-direction sweep; kind 4 applies 100 / 256) or w_l (second reverse sweep)
    };
    // MODE 5: tile t of a [neurons x samples] accumulator tile -> columns 32 t .. 32 t + 31 of this lane's sample row of signal array
    // `arr` (register i of lane (h, j): neuron 8 (i / 4) + 4 h + (i % 4), sample j), times `scale` (powers of two: exact)
    float sig_scale = 1.f;   // 1 / kappa once it is known (below)
    [[maybe_unused]] auto sig = [&](int arr, int t, const f32x16& y, float scale) {
        if constexpr (PG) {
            if (valid) {
                using f32x4 = float __attribute__((ext_vector_type(4)));
                float* row = a.sig + (size_t)arr * a.sig_pitch + (size_t)n * 256 + 32 * t + 4 * h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<f32x4*>(row + 8 * g4) = f32x4{y[4 * g4] * scale, y[4 * g4 + 1] * scale, y[4 * g4 + 2] * scale, y[4 * g4 + 3] * scale};
            }
        }
    };
    // to_regs, and the tile's value -> signal array `arr` (times 1 / kappa), the pre-data's tile -> `arr_pre` (as it is; < 0: none)
    [[maybe_unused]] auto to_regs_sig = [&](h8(&oh)[16], h8(&ol)[16], int arr, int arr_pre) {
        return [&oh, &ol, arr, arr_pre, &sig, &sig_scale, &park](auto T, EpiState& st, const auto& pd) {
            constexpr int t = decltype(T)::value;
            asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
            oh[2 * t] = st.hi[0];
            ol[2 * t] = st.lo[0];
            oh[2 * t + 1] = st.hi[1];
            ol[2 * t + 1] = st.lo[1];
            park(oh[2 * t], ol[2 * t], oh[2 * t + 1], ol[2 * t + 1]);
            if constexpr (PG) {
                sig(arr, t, st.vec(), sig_scale);
                if constexpr (!std::is_same_v<std::decay_t<decltype(pd)>, NoData>) {
                    if (arr_pre >= 0) sig(arr_pre, t, pd.v, 1.f);
                }
            }
            return NoData{};
        };
    };
    ws.stamp(10);   // (timing builds, tools/ts_report_adj.py: section starts 10 .. 15)
    // ---- seeds
    float gs = a.g_sdf[nn], gg[3], gr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gg[c] = a.g_grad[3 * nn + c];
        gr[c] = a.g_rgb[3 * nn + c];
    }
    if (!valid) {   // lanes beyond the end shadow the last sample: they must not contribute
        gs = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) gg[c] = gr[c] = 0.f;
    }
    float kappa = 1.f, inv_kappa = 1.f;
    {
        float m = fabsf(gs);
#pragma unroll
        for (int c = 0; c < 3; ++c) m = fmaxf(m, fmaxf(fabsf(gg[c]), fabsf(gr[c])));
        const unsigned e = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu;   // m in [2^(e-127), 2^(e-126))
        if (e >= 1u && e <= 253u) {
            kappa = __builtin_bit_cast(float, (254u - e) << 23);
            inv_kappa = __builtin_bit_cast(float, e << 23);
        }
    }
    sig_scale = inv_kappa;
    const float gsk = gs * kappa;
    float xb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xb[c] = kappa * gr[c] * rgb[c] * (1.f - rgb[c]);

    auto mask_of = [&](int c_slot) {
        return [&sh, c_slot](auto T, const char*) { return Act{sh.tile_load(c_slot, decltype(T)::value)}; };
    };
    auto pose = [&](int b, int i) { return uni ? rd<true>(Mu + 16 * b, i) : rd<false>(M + 16 * b, i); };

    // ---- colour lin4^T and the mask of c4: cb4 = (c4 > 0) * (W_c4^T xb)
    {
        const char* buf = ws.template acquire<0>();
        ws.template begin_c<HB_BWD>();
        ws.template pieces_all_c<HB_BWD>();
        static_for<8>([&](auto T) {
            constexpr int t = decltype(T)::value;
            const f32x16 c4 = sh.tile_load(HS_C + 3, t);
            const f32x16 w0 = tail_tile(buf, t, h), w1 = tail_tile(buf + TAIL_BYTES, t, h), w2 = tail_tile(buf + 2 * TAIL_BYTES, t, h);
            f32x16 v;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = c4[i] > 0.f ? fmaf(w0[i], xb[0], fmaf(w1[i], xb[1], w2[i] * xb[2])) : 0.f;
            split_tile(v, ah[2 * t], al[2 * t], ah[2 * t + 1], al[2 * t + 1]);
            sig(HSG_CB + 0, t, v, sig_scale);
            sig(HSG_C + 3, t, c4, 1.f);
        });
    }
    if constexpr (PG) {
        run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, ah, al, lane, h, mask_of(HS_C + 2), PhMask{}, to_regs_sig(bh, bl, HSG_CB + 1, HSG_C + 2), no_store);   // C3^T -> cb3
        run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, bh, bl, lane, h, mask_of(HS_C + 1), PhMask{}, to_regs_sig(ah, al, HSG_CB + 2, HSG_C + 1), no_store);   // C2^T -> cb2
        run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, ah, al, lane, h, mask_of(HS_C + 0), PhMask{}, to_regs_sig(bh, bl, HSG_CB + 3, HSG_C + 0), no_store);   // C1^T -> cb1
    } else {
    run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, ah, al, lane, h, mask_of(HS_C + 2), PhMask{}, to_regs(bh, bl), no_store);   // C3^T -> cb3
    run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, bh, bl, lane, h, mask_of(HS_C + 1), PhMask{}, to_regs(ah, al), no_store);   // C2^T -> cb2
    run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, ah, al, lane, h, mask_of(HS_C + 0), PhMask{}, to_regs(bh, bl), no_store);   // C1^T -> cb1
    }
    // ---- colour lin0^T, feature-vector rows -> fb (fragments, kept in the HS_FVEC slot for the W8 product)
    run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(
        ws, bh, bl, lane, h, no_pre, PhIdentity{},
        [&](auto T, EpiState& st, const auto&) {
            constexpr int t = decltype(T)::value;
            sh.frag_store(HS_FVEC * SLOT_BYTES, 2 * t, st.hi[0], st.lo[0]);
            sh.frag_store(HS_FVEC * SLOT_BYTES, 2 * t + 1, st.hi[1], st.lo[1]);
            sig(HSG_FB, t, st.vec(), sig_scale);
            return NoData{};
        },
        no_store);
    // ---- colour lin0^T, enc(g) slots (one tile) -> gb = g_grad + J_enc^T (.)
    float gb[3] = {0.f, 0.f, 0.f};
    {
        const char* buf = ws.template acquire<0>();
        ws.template begin_c<HB_BWD>();
        f32x16 m1 = zero16(), m2 = zero16();
        mma_tile<16, 0, HB_BWD>(ws, buf, bh, bl, m1, m2, lane);
        const f32x16 Mg = combine(m1, m2);
        float f[2][8];
        encode_v4h(g, h, f);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float fr = (float)(1 << (jj & 3));
            gb[jj >> 2] = fmaf(Mg[jj], (h ? -fr : fr) * other_half(f[0][jj], h), gb[jj >> 2]);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float fr = (float)(1 << jj);
            gb[2] = fmaf(Mg[8 + jj], (h ? -fr : fr) * other_half(f[1][jj], h), gb[2]);
        }
        gb[0] += h ? 0.f : Mg[12];
        gb[2] += h ? Mg[12] : 0.f;
        gb[1] += h ? 0.f : Mg[13];
#pragma unroll
        for (int c = 0; c < 3; ++c) gb[c] = fmaf(kappa, gg[c], half_sum(gb[c]));
        if constexpr (PG) {
            if (valid && h == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) a.gb_out[3 * (size_t)n + c] = gb[c] * inv_kappa;
            }
        }
    }
    // the rows of an X-space adjoint that belong to the leftover block: 2 tiles; register 8 (u & 1) + jj of tile u >> 1
    // <-> bone 8 u + jj.  Parked per bone (one float per lane) for the bone loop that follows.
    auto park_leftover = [&](const f32x16& La, const f32x16& Lb) {
        static_for<N_BONES>([&](auto B_) {
            constexpr int b = decltype(B_)::value;
            sh.f32_store(LEFTX + b * 256, b < 16 ? La[b] : Lb[b - 16]);
        });
    };
    // this lane's share of the leftover pair (r_1 | r_2) h of bone b in the sums of a row G
    auto add_leftover = [&](BoneSums& S, const Bone2& bn, int b) {
        const float Gl = sh.f32_load(LEFTX + b * 256);
        S.T0 = fmaf(Gl, (h ? bn.r[2] : bn.r[1]) * bn.hh, S.T0);
        S.T1[2] += h ? 0.f : Gl * bn.hh;
        S.T1[3] += h ? Gl * bn.hh : 0.f;
    };
    // own[s][jj]: this lane's stored features of the bone staged in LDS
    auto staged_features = [&](float(&own)[4][8]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const h8 fh = *reinterpret_cast<const h8*>(stage + (2 * s) * 1024 + lane * 16);
            const h8 fl = *reinterpret_cast<const h8*>(stage + (2 * s + 1) * 1024 + lane * 16);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) own[s][jj] = unsplit(fh[jj], fl[jj]);
        }
    };
    auto stage_bone = [&](int b) {   // stash -> LDS by DMA; lands under the MFMAs that follow, the next acquire covers it
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sh.rsrc, (lds_void_t*)(stage + i * 1024), 16, lane_x16(),
                                                     FEAT + (4 * b + (i >> 1)) * KS_BYTES + (i & 1) * 1024, 0, STASH_AUX);
    };

    ws.stamp(11);
    // ---- colour lin0^T over the feature rows (pass A of the input map): leftover rows first, then bone by bone
    //      (2 chunks each); per bone the colour network's share of qbar goes to the stash
    {
        f32x16 L1[2], L2[2];
        static_for<2>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const char* buf = ws.template acquire<0>();
            ws.template begin_c<HB_BWD>();
            L1[u] = zero16();
            L2[u] = zero16();
            mma_tile<16, 0, HB_BWD>(ws, buf, bh, bl, L1[u], L2[u], lane);
        });
        park_leftover(combine(L1[0], L2[0]), combine(L1[1], L2[1]));
        const int jbase = ws.goff - HB_BWD;   // stream offset of bone 0's first chunk (in flight)
        unsigned rem = nzw & ~1u;
        int b = 0;
#pragma unroll 1
        while (b < N_BONES) {
            const int nb = rem ? __builtin_ctz(rem) : N_BONES;
            rem &= rem - 1u;
            f32x16 G1[2], G2[2];
            const bool live = (nz >> b) & 1u;
            static_for<2>([&](auto U) {
                constexpr int u = decltype(U)::value;
                const char* buf = ws.template acquire<0>();
                if constexpr (u == 1) ws.goff = jbase + nb * (2 * HB_BWD);
                static_assert(HB_BONE == HB_BWD, "one size");
                ws.template begin_c<HB_BWD>();   // after the last bone: lin0 of the forward-direction sweep (a bone chunk, the same size)
                if (u == 0 && live) stage_bone(b);
                G1[u] = zero16();
                G2[u] = zero16();
                mma_tile<16, 0, HB_BWD>(ws, buf, bh, bl, G1[u], G2[u], lane);
            });
            if (live) {
                // the staged features were requested before tile 0's MFMAs; the acquire of tile 1 waited for them
                const Bone2 bn = coords(b);
                const float kk = -TAU2 * (1.f - bn.hh);
                float own[4][8];
                staged_features(own);
                BoneSums S;
                bone_sums<false>(combine(G1[0], G2[0]), combine(G1[1], G2[1]), own, bn.hh, h, S);
                add_leftover(S, bn, b);
                sums_reduce(S, false);
                float dq[3];
                dq_from_sums(S, bn, kk, dq);
#pragma unroll
                for (int c = 0; c < 3; ++c) sh.f32_store(QA + (3 * b + c) * 256, dq[c]);
            }
            b = nb;
        }
    }

    ws.stamp(12);
    // ---- J gb as fragments (the layout of the features): d(phi h)/dq . (R_b gb) per slot
    {
#pragma unroll 1
        for (int b = 0; b < N_BONES; ++b) {
            float left = 0.f;
            if ((nz >> b) & 1u) {
                // the bone's four feature blocks first, all in flight together and under the coordinate arithmetic below
                // (loaded one by one at their use they were 84 serialised HBM round trips per tile: 187 000 cycles in the
                // in-kernel stamps)
                h8 ffh[4], ffl[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) sh.frag_load(FEAT, 4 * b + s, ffh[s], ffl[s]);
                const Bone2 bn = coords(b);
                const float kk = -TAU2 * (1.f - bn.hh);
                float w[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) w[i] = pose(b, 4 * i) * gb[0] + pose(b, 4 * i + 1) * gb[1] + pose(b, 4 * i + 2) * gb[2];
                const float rw = bn.r[0] * w[0] + bn.r[1] * w[1] + bn.r[2] * w[2];
                const float dy[4] = {rw, (w[0] - bn.r[0] * rw) / bn.v, (w[1] - bn.r[1] * rw) / bn.v, (w[2] - bn.r[2] * rw) / bn.v};
                const float kr = kk * rw;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    h8 fh = ffh[s], fl = ffl[s];
                    float jg[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const float own = unsplit(fh[jj], fl[jj]);
                        if (s == 3 && jj == 7) {
                            jg[jj] = fmaf(bn.hh, h ? dy[1] : dy[0], own * kr);
                        } else {
                            const float fr = (float)(1 << slot_freq(s, jj));
                            jg[jj] = fmaf((h ? -fr : fr) * other_half(own, h), dy[slot_var(s, jj)], own * kr);
                        }
                    }
                    split8(jg, fh, fl);
                    sh.frag_store(GXB, 4 * b + s, fh, fl);
                }
                left = fmaf(bn.hh, h ? dy[3] : dy[2], (h ? bn.r[2] : bn.r[1]) * bn.hh * kr);
            }
            sh.f32_store(LEFT2 + b * 256, left);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            float f[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) f[jj] = (8 * u + jj < N_BONES) ? sh.f32_load(LEFT2 + (8 * u + jj) * 256) : 0.f;
            h8 fh, fl;
            split8(f, fh, fl);
            sh.frag_store(GXB, FEAT_BLOCKS + u, fh, fl);
        }
    }

    ws.stamp(13);
    // ---- forward-direction sweep
    auto pre4 = [&](int act_slot, int dz_slot) {
        return [&sh, act_slot, dz_slot](auto T, const char*) {
            constexpr int t = decltype(T)::value;
            return Act2{sh.tile_load(act_slot, t), sh.tile_load(dz_slot, t)};   // (x = dz_l; kind 4 applies its 100 / 256)
        };
    };
    auto fin4 = [&](h8(&oh)[16], h8(&ol)[16], int w_slot) {
        return [&oh, &ol, w_slot, &sh, &park, &sig, &sig_scale](auto T, EpiState& st, const auto& pd) {
            constexpr int t = decltype(T)::value;
            if constexpr (PG) {   // layer l = w_slot - HS_DZ: v_l, a_{l+1}, dz_l
                sig(HSG_V + (w_slot - HS_DZ), t, st.vec(), sig_scale);
                sig(HSG_A + (w_slot - HS_DZ), t, pd.v, 1.f);
                sig(HSG_DZ + (w_slot - HS_DZ), t, pd.x, BWD_INV);
            }
            asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
            oh[2 * t] = st.hi[0];
            ol[2 * t] = st.lo[0];
            oh[2 * t + 1] = st.hi[1];
            ol[2 * t + 1] = st.lo[1];
            park(oh[2 * t], ol[2 * t], oh[2 * t + 1], ol[2 * t + 1]);
            sh.tile_store(w_slot, t, st.wvec());
            return NoData{};
        };
    };
    // epilogue of a finished block with per-tile side data (block_epilogue with a `pre`)
    auto block_epilogue_pd = [&](auto& c1, auto& c2, auto&& ph, auto&& pre, auto&& fin) {
        static_for<8>([&](auto TI) {
            constexpr int ti = decltype(TI)::value;
            EpiState st;
            arm(st);
            st.c1 = c1[ti];
            st.c2 = c2[ti];
            const Act2 pd = pre(TI, (const char*)nullptr);
            Epi<true, std::remove_reference_t<decltype(ph)>, Act2> epi{st, ph, pd};
            epi.run_all();
            split_finish<true>(st);
            fin(TI, st, pd);
        });
    };
    feat_base = GXB;
    {   // lin0: J gb -> dzb0
        f32x16 c1[8], c2[8];
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
            c1[ti] = zero16();
            c2[ti] = zero16();
        }
        feature_pass(I2{}, BFalse{}, c1, c2, std::integral_constant<int, HB_LEFT>{}, std::integral_constant<int, HB_HID>{}, IP3{});
        block_epilogue_pd(c1, c2, PhFwdDir{}, pre4(HS_A1 + 0, HS_DZ + 0), fin4(ah, al, HS_DZ + 0));
    }
    run_layer_c<8, 16, 1, false, true, HB_HID, HB_HID>(ws, ah, al, lane, h, pre4(HS_A1 + 1, HS_DZ + 1), PhFwdDir{}, fin4(bh, bl, HS_DZ + 1), no_store);   // lin1
    run_layer_c<8, 16, 1, false, true, HB_HID, HB_HID>(ws, bh, bl, lane, h, pre4(HS_A1 + 2, HS_DZ + 2), PhFwdDir{}, fin4(ah, al, HS_DZ + 2), no_store);   // lin2
    run_layer_c<8, 16, 1, false, true, HB_HID, HB_HID>(ws, ah, al, lane, h, pre4(HS_A1 + 3, HS_DZ + 3), PhFwdDir{}, fin4(bh, bl, HS_DZ + 3), no_store);   // lin3
    {   // lin4 = [v3 | J gb] / sqrt2
        f32x16 c1[8], c2[8];
        static_for<8>([&](auto TI) {
            constexpr int ti = decltype(TI)::value;
            const char* buf = ws.template acquire<0>();
            constexpr int nbytes = ti < 7 ? HB_HID : HB_BONE;
            ws.template begin_c<nbytes>();
            c1[ti] = zero16();
            c2[ti] = zero16();
            mma_tile<16, 0, nbytes>(ws, buf, bh, bl, c1[ti], c2[ti], lane);
        });
        feature_pass(I2{}, BFalse{}, c1, c2, std::integral_constant<int, HB_LEFT>{}, std::integral_constant<int, HB_HID>{}, IP3{});
        block_epilogue_pd(c1, c2, PhFwdDir{}, pre4(HS_A1 + 4, HS_DZ + 4), fin4(ah, al, HS_DZ + 4));
    }
    feat_base = FEAT;
    run_layer_c<8, 16, 1, false, true, HB_HID, HB_HID>(ws, ah, al, lane, h, pre4(HS_A1 + 5, HS_DZ + 5), PhFwdDir{}, fin4(bh, bl, HS_DZ + 5), no_store);   // lin5
    run_layer_c<8, 16, 1, false, true, HB_HID, HB_HID>(ws, bh, bl, lane, h, pre4(HS_A1 + 6, HS_DZ + 6), PhFwdDir{}, fin4(ah, al, HS_DZ + 6), no_store);   // lin6
    run_layer_c<8, 16, 1, false, false, HB_HID, HB_HID>(ws, ah, al, lane, h, pre4(HS_A8, HS_DZ + 7), PhFwdDir{},                                       // lin7: only w_7
                                      [&](auto T, EpiState& st, const auto& pd) {
                                          if constexpr (PG) {
                                              sig(HSG_V + 7, decltype(T)::value, st.vec(), sig_scale);
                                              sig(HSG_A + 7, decltype(T)::value, pd.v, 1.f);
                                              sig(HSG_DZ + 7, decltype(T)::value, pd.x, BWD_INV);
                                          }
                                          sh.tile_store(HS_DZ + 7, decltype(T)::value, st.wvec());
                                          return NoData{};
                                      },
                                      no_store);

    ws.stamp(14);
    // ---- second reverse sweep.  ab_7 = W8[1:, :]^T fb + g_sdf W8[0, :];  zb_7 = sigma'_7 ab_7 + w_7
    auto pre5 = [&](int act_slot, int w_slot) {
        return [&sh, act_slot, w_slot](auto T, const char*) {
            constexpr int t = decltype(T)::value;
            return Act2{sh.tile_load(act_slot, t), sh.tile_load(w_slot, t)};
        };
    };
#pragma unroll
    for (int s = 0; s < 16; ++s) sh.frag_load(HS_FVEC * SLOT_BYTES, s, bh[s], bl[s]);
    run_layer_c<8, 16, 1, false, true, HB_HID, HB_BWD>(
        ws, bh, bl, lane, h,
        [&](auto T, const char* tail) {
            constexpr int t = decltype(T)::value;
            Act2 o{sh.tile_load(HS_A8, t), sh.tile_load(HS_DZ + 7, t)};
            const f32x16 w8 = tail_tile(tail, 0, h);
#pragma unroll
            for (int i = 0; i < 16; ++i) o.x[i] = fmaf(gsk * w8[i], dsoftplus_from_act(o.v[i]), o.x[i]);   // + sigma'_7 g_sdf W8[0, :]
            return o;
        },
        PhRev2{}, to_regs_sig(ah, al, HSG_ZB + 7, -1), no_store);
    run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, ah, al, lane, h, pre5(HS_A1 + 6, HS_DZ + 6), PhRev2{}, to_regs_sig(bh, bl, HSG_ZB + 6, -1), no_store);   // W7^T -> zb6
    run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, bh, bl, lane, h, pre5(HS_A1 + 5, HS_DZ + 5), PhRev2{}, to_regs_sig(ah, al, HSG_ZB + 5, -1), no_store);   // W6^T -> zb5
    run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, ah, al, lane, h, pre5(HS_A1 + 4, HS_DZ + 4), PhRev2{},                               // W5^T -> zb4 (kept)
                                     [&](auto T, EpiState& st, const auto&) {
                                         constexpr int t = decltype(T)::value;
                                         asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                                         bh[2 * t] = st.hi[0];
                                         bl[2 * t] = st.lo[0];
                                         bh[2 * t + 1] = st.hi[1];
                                         bl[2 * t + 1] = st.lo[1];
                                         sh.frag_store(HS_ZB4 * SLOT_BYTES, 2 * t, st.hi[0], st.lo[0]);
                                         sh.frag_store(HS_ZB4 * SLOT_BYTES, 2 * t + 1, st.hi[1], st.lo[1]);
                                         sig(HSG_ZB + 4, t, st.vec(), sig_scale);
                                         return NoData{};
                                     },
                                     no_store);
    run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, bh, bl, lane, h, pre5(HS_A1 + 3, HS_DZ + 3), PhRev2{}, to_regs_sig(ah, al, HSG_ZB + 3, -1), no_store);   // W4h^T -> zb3
    run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, ah, al, lane, h, pre5(HS_A1 + 2, HS_DZ + 2), PhRev2{}, to_regs_sig(bh, bl, HSG_ZB + 2, -1), no_store);   // W3^T -> zb2
    run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, bh, bl, lane, h, pre5(HS_A1 + 1, HS_DZ + 1), PhRev2{}, to_regs_sig(ah, al, HSG_ZB + 1, -1), no_store);   // W2^T -> zb1
    run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD>(ws, ah, al, lane, h, pre5(HS_A1 + 0, HS_DZ + 0), PhRev2{}, to_regs_sig(bh, bl, HSG_ZB + 0, -1), no_store);   // W1^T -> zb0

    ws.stamp(15);
    // ---- input map (pass B): X-adjoint rows W0^T zb0 + W4x^T zb4, leftover rows first, then bone by bone; per bone the
    //      pull, the Hessian-vector term and the pose gradients
    float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) sh.frag_load(HS_ZB4 * SLOT_BYTES, s, ah[s], al[s]);
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" : "+a"(bh[s]), "+a"(bl[s]), "+a"(ah[s]), "+a"(al[s]));
    {
        f32x16 L1[2], L2[2];
        static_for<2>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const char* buf0 = ws.template acquire<0>();
            ws.template begin_c<HB_BWD>();
            L1[u] = zero16();
            L2[u] = zero16();
            mma_tile<16, 0, HB_BWD>(ws, buf0, bh, bl, L1[u], L2[u], lane);
            const char* buf4 = ws.template acquire<0>();
            ws.template begin_c<HB_BWD>();
            mma_tile<16, 0, HB_BWD>(ws, buf4, ah, al, L1[u], L2[u], lane);
        });
        park_leftover(combine(L1[0], L2[0]), combine(L1[1], L2[1]));
        const int jbase = ws.goff - HB_BWD;   // stream offset of bone 0's first chunk (in flight)
        unsigned rem = nzw & ~1u;
        int b = 0;
#pragma unroll 1
        while (b < N_BONES) {
            const int nb = rem ? __builtin_ctz(rem) : N_BONES;
            rem &= rem - 1u;
            f32x16 G1[2], G2[2];
            const bool live = (nz >> b) & 1u;
            // the bone's 14 per-lane scalars of the tape (sums of the forward pass, the colour net's share of qbar, its
            // leftover row): requested here, four chunks ahead of their use -- at the use they were one more HBM round trip
            // per bone in front of 1 500 dependent instructions
            BoneSums X;
            float qa[3], Gl_b = 0.f;
            X.T0 = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) X.T1[q4] = X.T2[q4] = 0.f;
            qa[0] = qa[1] = qa[2] = 0.f;
            if (live) {
                X.T0 = sh.f32_load(TS + (9 * b) * 256);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    X.T1[q4] = sh.f32_load(TS + (9 * b + 1 + q4) * 256);
                    X.T2[q4] = sh.f32_load(TS + (9 * b + 5 + q4) * 256);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) qa[c] = sh.f32_load(QA + (3 * b + c) * 256);
                Gl_b = sh.f32_load(LEFTX + b * 256);
            }
            static_for<2>([&](auto U) {
                constexpr int u = decltype(U)::value;
                const char* buf0 = ws.template acquire<0>();
                ws.template begin_c<HB_BWD>();
                if (u == 0 && live) stage_bone(b);
                G1[u] = zero16();
                G2[u] = zero16();
                mma_tile<16, 0, HB_BWD>(ws, buf0, bh, bl, G1[u], G2[u], lane);
                const char* buf4 = ws.template acquire<0>();
                if constexpr (u == 1) ws.goff = jbase + nb * (4 * HB_BWD);
                if constexpr (u == 1) {   // after the last bone: the next tile's first chunk (the one size that is not a constant)
                    ws.begin(nb == N_BONES ? (more ? FIRST_CHUNK : 0) : HB_BWD);
                    mma_tile<16, 0, 1>(ws, buf4, ah, al, G1[u], G2[u], lane);
                } else {
                    ws.template begin_c<HB_BWD>();
                    mma_tile<16, 0, HB_BWD>(ws, buf4, ah, al, G1[u], G2[u], lane);
                }
            });
            if (live) {
                const Bone2 bn = coords(b);
                const float sg = 1.f - bn.hh;
                const float kk = -TAU2 * sg;
                const float k2 = -TAU2 * TAU2 * sg * (2.f * bn.hh - 1.f);
                float own[4][8];
                staged_features(own);
                BoneSums S;
                bone_sums<false>(combine(G1[0], G2[0]), combine(G1[1], G2[1]), own, bn.hh, h, S);
                S.T0 = fmaf(Gl_b, (h ? bn.r[2] : bn.r[1]) * bn.hh, S.T0);   // the leftover pair's share (add_leftover)
                S.T1[2] += h ? 0.f : Gl_b * bn.hh;
                S.T1[3] += h ? Gl_b * bn.hh : 0.f;
                sums_reduce(S, false);
                float dqB[3];
                dq_from_sums(S, bn, kk, dqB);
                // d sdf / d features of the forward pass (the tape, X): its own d/dq and its Hessian-vector product along R_b gb
                float dqX[3], hv[3], w[3];
                dq_from_sums(X, bn, kk, dqX);
#pragma unroll
                for (int i = 0; i < 3; ++i) w[i] = pose(b, 4 * i) * gb[0] + pose(b, 4 * i + 1) * gb[1] + pose(b, 4 * i + 2) * gb[2];
                hv_from_sums(X, bn, kk, k2, w, hv);
                float qbar[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) qbar[c] = (dqB[c] + hv[c] + qa[c]) * inv_kappa;
#pragma unroll
                for (int c = 0; c < 3; ++c) gp[c] += pose(b, c) * qbar[0] + pose(b, 4 + c) * qbar[1] + pose(b, 8 + c) * qbar[2];
                if (a.g_bt_inv != nullptr || a.g_T_pose != nullptr) {
                    // g_bt_inv[b][i][:3] += qbar_i p + dqX_i gb;  [i][3] += qbar_i;  g_T_pose[b][i] -= qbar_i
                    float vals[12];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) vals[4 * i + c] = fmaf(qbar[i], p[c], dqX[i] * gb[c] * inv_kappa);
                        vals[4 * i + 3] = qbar[i];
                    }
                    // A sample within ~2 mm of a bone's local origin has a true gradient ~1e6 x the others' (1 / v^2 terms of the bone
                    // map): beyond the fp16 fragments' range however the seeds are scaled, its adjoint quantities overflow and come out
                    // as inf / NaN.  Such a sample is DROPPED from the pose gradients (and gets g_pts = 0) instead of poisoning every
                    // pose leaf with a NaN; the fp32 reference would add a huge finite value there.
                    bool finite = true;
#pragma unroll
                    for (int k = 0; k < 12; ++k) finite = finite && fabsf(vals[k]) <= 3.0e38f;
                    const bool mine = valid && h == 0 && finite;   // both halves hold the same values: one of them contributes
                    if (uni) {
#pragma unroll
                        for (int k = 0; k < 12; ++k) vals[k] = wave_sum64(mine ? vals[k] : 0.f);
                        if (lane == 0 && a.pose_part != nullptr) {   // this wave's row of its frame for this tile (LDS), added up in a fixed order afterwards
#pragma unroll
                            for (int k = 0; k < 12; ++k) prow[b * 12 + k] += vals[k];
                        } else if (lane == 0) {
                            if (a.g_bt_inv != nullptr) {
                                float* gm = a.g_bt_inv + ((size_t)frame0 * N_BONES + b) * 16;
#pragma unroll
                                for (int k = 0; k < 12; ++k) atomicAdd(gm + k, vals[k]);
                            }
                            if (a.g_T_pose != nullptr) {
                                float* gt = a.g_T_pose + ((size_t)frame0 * N_BONES + b) * 3;
#pragma unroll
                                for (int i = 0; i < 3; ++i) atomicAdd(gt + i, -vals[4 * i + 3]);
                            }
                        }
                    } else if (a.pose_part != nullptr) {
                        // a wave that holds samples of TWO frames -- its first sample's and the last lane's (a dense list crossing a
                        // frame boundary: the next frame; the wave of a compact list that holds the stand-in, whose frame is its own
                        // dense index's: any; lanes past the list replicate the last sample) -- : the same sums per frame, the other
                        // frame's lanes contributing zeros; row 0: the first sample's frame, row 1: the other one
                        const int f_b = __builtin_amdgcn_readlane(frame, 63);
                        for (int sl = 0; sl < POSE_FRAMES; ++sl) {
                            const int f = sl == 0 ? frame0 : f_b;
#pragma unroll
                            for (int k = 0; k < 12; ++k) {
                                const float sk = wave_sum64((mine && frame == f) ? vals[k] : 0.f);
                                if (lane == 0) prow[sl * POSE_ROW + b * 12 + k] += sk;
                            }
                        }
                    } else if (mine) {
                        if (a.g_bt_inv != nullptr) {
                            float* gm = a.g_bt_inv + ((size_t)frame * N_BONES + b) * 16;
#pragma unroll
                            for (int k = 0; k < 12; ++k) atomicAdd(gm + k, vals[k]);
                        }
                        if (a.g_T_pose != nullptr) {
                            float* gt = a.g_T_pose + ((size_t)frame * N_BONES + b) * 3;
#pragma unroll
                            for (int i = 0; i < 3; ++i) atomicAdd(gt + i, -vals[4 * i + 3]);
                        }
                    }
                }
            }
            b = nb;
        }
    }
    if (valid && h == 0) {
        const bool finite = fabsf(gp[0]) <= 3.0e38f && fabsf(gp[1]) <= 3.0e38f && fabsf(gp[2]) <= 3.0e38f;   // (see the pose gradients above)
        a.g_pts[3 * n] = finite ? gp[0] : 0.f;
        a.g_pts[3 * n + 1] = finite ? gp[1] : 0.f;
        a.g_pts[3 * n + 2] = finite ? gp[2] : 0.f;
        if (!finite) atomicAdd(&g_hn_dropped_samples, 1ull);   // (hn_dropped_samples: how often this happened)
    }
    if constexpr (PG) {
        // ... and from the parameter gradients: the rows this lane wrote into the signal arrays hold inf / NaN from the forward-direction
        // sweep on (J gb left the fragments' range); one such row would turn every parameter gradient into NaN.  Zeros instead: the sample
        // does not contribute (the fp32 launch sequence adds its huge finite share).
        const bool finite = fabsf(gp[0]) <= 3.0e38f && fabsf(gp[1]) <= 3.0e38f && fabsf(gp[2]) <= 3.0e38f;
        if (valid && !finite) {
            using f32x4 = float __attribute__((ext_vector_type(4)));
#pragma unroll 1
            for (int arr = 0; arr < HSG_COUNT; ++arr) {
                float* row = a.sig + (size_t)arr * a.sig_pitch + (size_t)n * 256 + 4 * h;
#pragma unroll 1
                for (int q = 0; q < 32; ++q) *reinterpret_cast<f32x4*>(row + 8 * q) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (h == 0) a.gb_out[3 * (size_t)n] = a.gb_out[3 * (size_t)n + 1] = a.gb_out[3 * (size_t)n + 2] = 0.f;
        }
    }
}
