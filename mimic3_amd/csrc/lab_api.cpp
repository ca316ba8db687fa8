// lab_api.cpp — the hooks of include/mi355vits_lab.h: kernel unit tests, a conv micro-benchmark, the box probes, the MFMA layout
// self test.  NOT part of the product library (libmi355vits.so exports include/mi355vits.h only): linked into
//   libmi355vits_hooks.so   the product's own objects + this file (the kernel-level GPU tests run the product's kernels through it),
//   libmi355vits_lab.so     the -DMI355_LAB build (A/B switches, tools/), and the CPU model of the kernels (tests/emu).
#include "c_api_internal.h"
#include "mi355vits_lab.h"

extern "C" {

int mi355vits_probe_weights(mi355vits_handle h, double out[8]) {
    if (!h || !out) return MI355VITS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->eng->mu);
    return guarded(h, [&] { h->eng->probe_weights(out); });
}

// kernel unit-test hooks (host buffers in, host buffers out)
int mi355vits_test_conv1d(int device, const mi355vits_conv_test* t) {
    return guarded(nullptr, [&] {
        if (!t || !t->x || !t->w || !t->y) throw EngineError(MI355VITS_ERR_INVALID, "null argument");
        HIP_CHECK(hipSetDevice(device));
        const size_t nx = (size_t)t->B * t->Cin * t->T, ny = (size_t)t->B * t->Cout * t->T;
        const size_t nw = (size_t)t->Cout * t->Cin * t->K;
        DevBuf dx(nx * 4), dy(ny * 4), dw(nw * 4), db(t->Cout * 4), dres(ny * 4), dil(t->B * 4), dol(t->B * 4);
        std::vector<float> packed;
        HIP_CHECK(hipMemcpy(dx.p, t->x, nx * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dy.p, t->y, ny * 4, hipMemcpyHostToDevice));
        ConvArgs a;
        a.x = dx.as<float>(); a.x_bs = (long)t->Cin * t->T; a.x_ld = t->T;
        a.y = dy.as<float>(); a.y_bs = (long)t->Cout * t->T; a.y_ld = t->T;
        a.B = t->B; a.Cin = t->Cin; a.Cout = t->Cout; a.T = t->T; a.K = t->K; a.dil = t->dilation;
        a.pad = (t->K * t->dilation - t->dilation) / 2;
        a.in_slope = t->in_slope; a.relu = t->relu; a.out_scale = t->out_scale; a.res_sub = t->res_sub;
        a.accumulate = t->accumulate;
        if (t->bias) { HIP_CHECK(hipMemcpy(db.p, t->bias, t->Cout * 4, hipMemcpyHostToDevice)); a.bias = db.as<float>(); }
        if (t->res) {
            HIP_CHECK(hipMemcpy(dres.p, t->res, ny * 4, hipMemcpyHostToDevice));
            a.res = dres.as<float>(); a.res_bs = a.y_bs; a.res_ld = t->T;
        }
        if (t->in_len) { HIP_CHECK(hipMemcpy(dil.p, t->in_len, t->B * 4, hipMemcpyHostToDevice)); a.in_len = dil.as<int>(); }
        if (t->out_len) { HIP_CHECK(hipMemcpy(dol.p, t->out_len, t->B * 4, hipMemcpyHostToDevice)); a.out_len = dol.as<int>(); }
        if (t->impl == 3) {  // the encoder's slice kernel (k_enc_b3); split convs: the raw slice sums added up here, no epilogue
            if (!enc_conv_b3_supported(t->Cin, t->Cout, t->K, t->dilation)) throw EngineError(MI355VITS_ERR_INVALID, "shape not supported by the encoder slice kernel");
            std::vector<uint32_t> b3(bf16x3_packed_words_mode(t->Cout, t->Cin, t->K, EPI_STD));
            pack_conv_weights_bf16x3_mode(t->w, t->Cout, t->Cin, t->K, EPI_STD, 1, b3.data());
            DevBuf db3(b3.size() * 4);
            HIP_CHECK(hipMemcpy(db3.p, b3.data(), b3.size() * 4, hipMemcpyHostToDevice));
            a.wb3 = db3.as<float>();
            a.math = MATH_BF16X3;
            a.ksplit = enc_conv_b3_slices(t->Cin);
            DevBuf dpart(ny * 4 * (size_t)a.ksplit);
            if (a.ksplit > 1) {
                if (t->bias || t->res || t->relu || t->accumulate) throw EngineError(MI355VITS_ERR_INVALID, "split conv: raw sums only");
                a.part = dpart.as<float>();
            }
            launch_enc_conv_b3(a, nullptr);
            HIP_CHECK(hipDeviceSynchronize());
            HIP_CHECK(hipGetLastError());
            if (a.ksplit > 1) {
                std::vector<float> parts(ny * (size_t)a.ksplit);
                HIP_CHECK(hipMemcpy(parts.data(), dpart.p, parts.size() * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < ny; ++i) {
                    float v = parts[i];
                    for (int sl = 1; sl < a.ksplit; ++sl) v += parts[(size_t)sl * ny + i];
                    t->y[i] = v;
                }
                return;
            }
        } else if (t->impl == 4) {  // the 128-channel resblock conv with every input channel resident (k_rb_conv / k_rb_conv_pw)
            if (!rb_conv_supported(a)) throw EngineError(MI355VITS_ERR_INVALID, "shape not supported by the resident-input kernel");
            std::vector<uint32_t> pp(p16_packed_words(t->Cout, t->Cin, t->K));
            pack_conv_weights_p16(t->w, t->Cout, t->Cin, t->K, pp.data());
            DevBuf dpp(pp.size() * 4);
            HIP_CHECK(hipMemcpy(dpp.p, pp.data(), pp.size() * 4, hipMemcpyHostToDevice));
            a.w = dpp.as<float>();
            a.math = MATH_BF16X3;
            a.in_len_host = t->in_len;
            launch_rb_conv(a, nullptr);
            HIP_CHECK(hipDeviceSynchronize());
        } else if (t->impl == 1 || t->impl == 2) {
            if (!conv1d_mfma_supported(t->Cin, t->Cout, t->K, t->dilation)) throw EngineError(MI355VITS_ERR_INVALID, "shape not supported by the MFMA kernel");
            packed.resize(mfma_packed_floats(t->Cout, t->Cin, t->K));
            pack_conv_weights_mfma(t->w, t->Cout, t->Cin, t->K, packed.data());
            DevBuf dp(packed.size() * 4);
            HIP_CHECK(hipMemcpy(dp.p, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
            a.w = dp.as<float>();
            std::vector<uint32_t> b3;
            std::unique_ptr<DevBuf> db3;
            if (t->impl == 2) {  // split-bf16 staged kernel (MATH_BF16X3)
                if (!conv1d_b3_supported(t->Cin, t->Cout, t->K, t->dilation, t->T)) throw EngineError(MI355VITS_ERR_INVALID, "shape not supported by the split-bf16 kernel");
                b3.resize(bf16x3_packed_words_mode(t->Cout, t->Cin, t->K, EPI_STD));
                pack_conv_weights_bf16x3_mode(t->w, t->Cout, t->Cin, t->K, EPI_STD, 1, b3.data());
                db3.reset(new DevBuf(b3.size() * 4));
                HIP_CHECK(hipMemcpy(db3->p, b3.data(), b3.size() * 4, hipMemcpyHostToDevice));
                a.wb3 = db3->as<float>();
                a.math = MATH_BF16X3;
            }
            launch_conv1d_mfma(a, nullptr);
            HIP_CHECK(hipDeviceSynchronize());
        } else {
            HIP_CHECK(hipMemcpy(dw.p, t->w, nw * 4, hipMemcpyHostToDevice));
            a.w = dw.as<float>();
            launch_conv1d_generic(a, nullptr);
            HIP_CHECK(hipDeviceSynchronize());
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(t->y, dy.p, ny * 4, hipMemcpyDeviceToHost));
    });
}

int mi355vits_test_conv_transpose1d(int device, int impl, int B, int Cin, int Cout, int Tin, int K, int stride,
                                    const float* x, const float* w, const float* bias, float in_slope, float* y) {
    return guarded(nullptr, [&] {
        if (!x || !w || !y) throw EngineError(MI355VITS_ERR_INVALID, "null argument");
        HIP_CHECK(hipSetDevice(device));
        const size_t nx = (size_t)B * Cin * Tin, ny = (size_t)B * Cout * Tin * stride, nw = (size_t)Cin * Cout * K;
        DevBuf dx(nx * 4), dy(ny * 4), dw(nw * 4), db(Cout * 4);
        HIP_CHECK(hipMemcpy(dx.p, x, nx * 4, hipMemcpyHostToDevice));
        if (impl == 3) {
            // the resident-input polyphase kernels (k_ups_pl: 256 -> 128, 128 -> 64; k_ups64: 64 -> 32), MATH_BF16X3; a valid length per
            // row is part of their contract (the engine always has one): every row at full length here
            const int taps = convt_taps(K, stride);
            std::vector<float> wv((size_t)stride * Cout * Cin * taps), bv((size_t)stride * Cout);
            convt_to_polyphase(w, bias, Cin, Cout, K, stride, wv.data(), bv.data());
            std::vector<uint32_t> pp(p16_packed_words(stride * Cout, Cin, taps));
            pack_conv_weights_p16n(wv.data(), stride * Cout, Cin, taps, pp.data());
            std::vector<int> lens((size_t)B, Tin);
            DevBuf dpp(pp.size() * 4), dbias(bv.size() * 4), dlen((size_t)B * 4);
            HIP_CHECK(hipMemcpy(dpp.p, pp.data(), pp.size() * 4, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(dbias.p, bv.data(), bv.size() * 4, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(dlen.p, lens.data(), (size_t)B * 4, hipMemcpyHostToDevice));
            ConvArgs u;
            u.x = dx.as<float>(); u.x_bs = (long)Cin * Tin; u.x_ld = Tin;
            u.y = dy.as<float>(); u.y_bs = (long)Cout * Tin * stride; u.y_ld = Tin * stride;
            u.w = dpp.as<float>(); u.bias = dbias.as<float>(); u.in_len = dlen.as<int>(); u.in_len_host = lens.data();
            u.Cin = Cin; u.Cout = stride * Cout; u.K = taps; u.dil = 1;
            u.in_slope = in_slope; u.pad = taps - 1; u.Tin = Tin;
            u.shuf_s = stride; u.shuf_p = (K - stride) / 2; u.shuf_cout = Cout; u.shuf_T = Tin * stride;
            u.B = B; u.T = Tin + taps - 1;
            u.math = MATH_BF16X3;
            if (!ups_pl_supported(u)) throw EngineError(MI355VITS_ERR_INVALID, "shape not supported by the resident-input polyphase kernels");
            launch_ups_pl(u, nullptr);
            HIP_CHECK(hipDeviceSynchronize());
        } else if (impl == 1 || impl == 2) {
            // polyphase filters on the MFMA conv kernel (the path Engine uses); impl 2: the split-bf16 staged kernels
            // (MATH_BF16X3; 64-channel chunks run the persistent producer / consumer form)
            const int taps = convt_taps(K, stride);
            std::vector<float> wv((size_t)stride * Cout * Cin * taps), bv((size_t)stride * Cout);
            convt_to_polyphase(w, bias, Cin, Cout, K, stride, wv.data(), bv.data());
            if (!conv1d_mfma_supported(Cin, stride * Cout, taps, 1)) throw EngineError(MI355VITS_ERR_INVALID, "shape not supported by the MFMA kernel");
            std::vector<float> pk(mfma_packed_floats(stride * Cout, Cin, taps));
            pack_conv_weights_mfma(wv.data(), stride * Cout, Cin, taps, pk.data());
            DevBuf dp(pk.size() * 4), dbias(bv.size() * 4);
            HIP_CHECK(hipMemcpy(dp.p, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(dbias.p, bv.data(), bv.size() * 4, hipMemcpyHostToDevice));
            ConvArgs u;
            u.x = dx.as<float>(); u.x_bs = (long)Cin * Tin; u.x_ld = Tin;
            u.y = dy.as<float>(); u.y_bs = (long)Cout * Tin * stride; u.y_ld = Tin * stride;
            u.w = dp.as<float>(); u.bias = dbias.as<float>();
            u.Cin = Cin; u.Cout = stride * Cout; u.K = taps; u.dil = 1;
            u.in_slope = in_slope; u.pad = taps - 1; u.Tin = Tin;
            u.shuf_s = stride; u.shuf_p = (K - stride) / 2; u.shuf_cout = Cout; u.shuf_T = Tin * stride;
            u.B = B; u.T = Tin + taps - 1;
            std::vector<uint32_t> b3;
            std::unique_ptr<DevBuf> db3;
            if (impl == 2) {
                if (!conv1d_b3_supported(Cin, stride * Cout, taps, 1, 1 << 30)) throw EngineError(MI355VITS_ERR_INVALID, "shape not supported by the split-bf16 kernel");
                b3.resize(bf16x3_packed_words_mode(stride * Cout, Cin, taps, EPI_STD));
                pack_conv_weights_bf16x3_mode(wv.data(), stride * Cout, Cin, taps, EPI_STD, 1, b3.data());
                db3.reset(new DevBuf(b3.size() * 4));
                HIP_CHECK(hipMemcpy(db3->p, b3.data(), b3.size() * 4, hipMemcpyHostToDevice));
                u.wb3 = db3->as<float>();
                u.math = MATH_BF16X3;
                u.fixed_rule = 1;
            }
            launch_conv1d_mfma(u, nullptr);
            HIP_CHECK(hipDeviceSynchronize());
        } else {
            HIP_CHECK(hipMemcpy(dw.p, w, nw * 4, hipMemcpyHostToDevice));
            ConvTArgs a;
            a.x = dx.as<float>(); a.x_bs = (long)Cin * Tin; a.x_ld = Tin;
            a.y = dy.as<float>(); a.y_bs = (long)Cout * Tin * stride; a.y_ld = Tin * stride;
            a.w = dw.as<float>();
            if (bias) { HIP_CHECK(hipMemcpy(db.p, bias, Cout * 4, hipMemcpyHostToDevice)); a.bias = db.as<float>(); }
            a.B = B; a.Cin = Cin; a.Cout = Cout; a.Tin = Tin; a.K = K; a.stride = stride; a.pad = (K - stride) / 2;
            a.in_slope = in_slope;
            launch_conv_transpose1d(a, nullptr);
            HIP_CHECK(hipDeviceSynchronize());
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(y, dy.p, ny * 4, hipMemcpyDeviceToHost));
    });
}

int mi355vits_bench_conv1d(int device, int B, int Cin, int Cout, int T, int K, int dilation, int epi, int reps,
                           float* ms_per_launch) {
    return guarded(nullptr, [&] {
        HIP_CHECK(hipSetDevice(device));
        if (!conv1d_mfma_supported(Cin, Cout, K, dilation) || reps < 1 || !ms_per_launch) throw EngineError(MI355VITS_ERR_INVALID, "bad arguments");
        const int H = Cout / 2;
        const size_t nx = (size_t)B * Cin * T, ny = (size_t)B * Cout * T, nw = (size_t)Cout * Cin * K;
        std::vector<float> hx(nx), hw(nw), pk(mfma_packed_floats(Cout, Cin, K));
        unsigned st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = rnd() * 0.05f;
        pack_conv_weights_mfma_mode(hw.data(), Cout, Cin, K, epi == 1 ? EPI_GATE : EPI_STD, pk.data());
        DevBuf dx(nx * 4), dy(ny * 4), dy2(ny * 4), dres(ny * 4), dp(pk.size() * 4), db(Cout * 4);
        HIP_CHECK(hipMemcpy(dx.p, hx.data(), nx * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dp.p, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemset(dy.p, 0, ny * 4));
        HIP_CHECK(hipMemset(dy2.p, 0, ny * 4));
        HIP_CHECK(hipMemset(dres.p, 0, ny * 4));
        HIP_CHECK(hipMemset(db.p, 0, Cout * 4));
        ConvArgs a;
        a.x = dx.as<float>(); a.x_bs = (long)Cin * T; a.x_ld = T;
        a.w = dp.as<float>(); a.bias = db.as<float>();
        a.B = B; a.Cin = Cin; a.Cout = Cout; a.T = T; a.K = K; a.dil = dilation; a.pad = (K * dilation - dilation) / 2;
        if (epi == 1) {
            a.epi = EPI_GATE; a.H = H;
            a.y = dy.as<float>(); a.y_bs = (long)H * T; a.y_ld = T;
        } else if (epi == 2) {
            a.epi = EPI_RESSKIP; a.H = Cout / 2;
            a.y = dy.as<float>(); a.y_bs = (long)a.H * T; a.y_ld = T;
            a.y2 = dy2.as<float>(); a.y2_bs = (long)a.H * T; a.y2_ld = T;
        } else {
            a.y = dy.as<float>(); a.y_bs = (long)Cout * T; a.y_ld = T;
            a.res = dres.as<float>(); a.res_bs = a.y_bs; a.res_ld = T;
            a.in_slope = 0.1f;
        }
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) launch_conv1d_mfma(a, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i) launch_conv1d_mfma(a, nullptr);
        HIP_CHECK(hipEventRecord(e1, nullptr));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *ms_per_launch = ms / reps;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    });
}

int mi355vits_probe_device(int device, double out[8]) {
    return guarded(nullptr, [&] {
        if (!out) throw EngineError(MI355VITS_ERR_INVALID, "bad arguments");
        HIP_CHECK(hipSetDevice(device));
        const int cus = current_device_cu_count();
        struct Ev {  // RAII: a HIP error below must not leak the events
            hipEvent_t e = nullptr;
            Ev() { HIP_CHECK(hipEventCreate(&e)); }
            ~Ev() { if (e) (void)hipEventDestroy(e); }
        } ev0, ev1;
        hipEvent_t e0 = ev0.e, e1 = ev1.e;
        auto timed = [&](auto&& launch, int warm, int reps) {
            for (int i = 0; i < warm; ++i) launch();
            HIP_CHECK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < reps; ++i) launch();
            HIP_CHECK(hipEventRecord(e1, nullptr));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0.0f;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            return (double)ms / reps;
        };
        // one zero-filled GiB serves everything: [0, 2.6 MB) the L2 table, [0, 24 MB) the big table, the halves as copy source / destination
        const size_t gib = (size_t)1 << 30;
        DevBuf mem(gib), sink((size_t)cus * 4 + 16);
        HIP_CHECK(hipMemset(mem.p, 0, gib));
        const int n16 = 256 * 8 * 80;  // 163,840 x 16 B = 2.62 MB
        const int reps_in = 8;
        const double ms_stream = timed([&] { launch_probe_l2_stream(mem.p, n16, reps_in, sink.as<unsigned>(), cus, nullptr); }, 1, 3);
        out[0] = (double)cus * reps_in * n16 * 16.0 / (ms_stream * 1e-3) / 1e9;
        const int steps = 2000;
        auto chase = [&](unsigned nlines, int warm) {
            const double ms = timed([&] { launch_probe_l2_latency(mem.as<unsigned>(), steps, nlines, sink.as<unsigned>(), cus, nullptr); }, warm, 2);
            return ms * 1e6 / steps;
        };
        out[1] = chase(1u << 15, 2);   // 2 MB: after two passes every XCD's L2 holds what its waves touch
        out[6] = chase(1u << 19, 2);   // 32 MB: past an L2, inside the memory-side cache
        out[7] = chase(1u << 24, 1);   // 1 GiB: HBM (and the TLB's reach)
        const long ncopy16 = (256L << 20) / 16;
        char* base = static_cast<char*>(mem.p);
        const double ms_copy = timed([&] { launch_probe_copy(base, base + gib / 2, ncopy16, cus * 16, nullptr); }, 1, 3);
        out[2] = 2.0 * ncopy16 * 16.0 / (ms_copy * 1e-3) / 1e9;
        out[3] = (double)cus;
        const long slice16 = ncopy16 / cus;
        const double ms_mixed = timed([&] { launch_probe_l2_mixed(mem.p, n16, reps_in, base + gib / 4, base + gib / 2, slice16, sink.as<unsigned>(), cus, nullptr); }, 1, 3);
        out[4] = (double)cus * reps_in * n16 * 16.0 / (ms_mixed * 1e-3) / 1e9;
        const int big16 = n16 * 9;
        const double ms_big = timed([&] { launch_probe_l2_stream(mem.p, big16, 1, sink.as<unsigned>(), cus, nullptr); }, 1, 3);
        out[5] = (double)cus * big16 * 16.0 / (ms_big * 1e-3) / 1e9;
    });
}

int mi355vits_test_mfma_layout(int device, float* err) {
    return guarded(nullptr, [&] {
        HIP_CHECK(hipSetDevice(device));
        DevBuf d((1024 + 256 + 256) * 4);
        launch_mfma_selftest(d.as<float>(), nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        std::vector<float> h(1536);
        HIP_CHECK(hipMemcpy(h.data(), d.p, 1536 * 4, hipMemcpyDeviceToHost));
        float worst = 0.0f;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float ref = 0;
                for (int k = 0; k < 2; ++k) ref += (float)(i + 100 * k + 1) * (float)(3 * j - 7 * k + 2);
                worst = std::max(worst, std::fabs(ref - h[i * 32 + j]));
            }
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float ref = 0;
                for (int k = 0; k < 4; ++k) ref += (float)(i + 100 * k + 1) * (float)(3 * j - 7 * k + 2);
                worst = std::max(worst, std::fabs(ref - h[1024 + i * 16 + j]));
            }
        for (int i = 0; i < 16; ++i)  // v_mfma_f32_16x16x32_bf16
            for (int j = 0; j < 16; ++j) {
                float ref = 0;
                for (int k = 0; k < 32; ++k) ref += (float)(i + 2 * k) * (float)(3 * j - k + 2);
                worst = std::max(worst, std::fabs(ref - h[1280 + i * 16 + j]));
            }
        if (err) *err = worst;
        if (worst > 1e-3f) throw EngineError(MI355VITS_ERR_INTERNAL, "MFMA fragment layout differs from the one the kernels assume");
    });
}

}  // extern "C"
