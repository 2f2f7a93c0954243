// Standalone timing harness for the Speller row kernels (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I automatic-speech-recognition_amd/csrc tools/micro/bench_rows.hip -o tools/micro/bench_rows
#include "gemm.hip"
#include "common.hip"
#include "speller.hip"
#include <vector>
#include <cstdio>
#include <cstring>

template <class T> static T* dalloc(size_t n, float fill = 0.01f) {
    T* p; hipMalloc(&p, n * sizeof(T));
    std::vector<T> h(n);
    for (size_t i = 0; i < n; ++i) {
        float v = fill * (float)((i * 2654435761u) % 2001) / 1000.f - fill;
        if (sizeof(T) == 2) { unsigned int u; memcpy(&u, &v, 4); h[i] = (T)(u >> 16); } else h[i] = (T)v;
    }
    hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice);
    return p;
}

int main() {
    const int B = 48, Tp = 160, Hd = 512, A = 128, D = 512, NL = 1, E = 128, V = 30, U = 191, G = 4;
    const int I0D = E + Hd + D, GD = G * D;
    DecDev d; memset(&d, 0, sizeof(d));
    d.B = B; d.Tp = Tp; d.Hd = Hd; d.A = A; d.D = D; d.NL = NL; d.E = E; d.V = V; d.U = U; d.mode = LAS_ATT_ADD; d.fb = 1.f;
    d.enc = dalloc<float>((size_t)B * Tp * Hd); d.keys = dalloc<float>((size_t)B * Tp * A);
    std::vector<int> hl(B, Tp - 5); int* len; hipMalloc(&len, B * 4); hipMemcpy(len, hl.data(), B * 4, hipMemcpyHostToDevice); d.enc_len = len;
    d.Ws = dalloc<float>((size_t)D * A); d.u = dalloc<float>(A); d.emb = dalloc<float>((size_t)V * E);
    d.Wv = dalloc<float>((size_t)D * V); d.bv = dalloc<float>(V);
    std::vector<int> ht((size_t)U * B, 3); int* tok; hipMalloc(&tok, U * B * 4); hipMemcpy(tok, ht.data(), U * B * 4, hipMemcpyHostToDevice); d.tok_in = tok;
    d.logits = dalloc<float>((size_t)U * B * V); d.alphas = dalloc<float>((size_t)U * B * Tp);
    d.hs = dalloc<float>((size_t)(U + 1) * B * D); d.cs = dalloc<float>((size_t)(U + 1) * B * D);
    d.gates = dalloc<float>((size_t)U * B * GD); d.xin0 = dalloc<float>((size_t)U * B * I0D);
    d.xbf = dalloc<unsigned short>((size_t)B * I0D); d.dgbf = dalloc<unsigned short>((size_t)B * GD);
    d.Wsbf = dalloc<unsigned short>((size_t)D * A); d.keysbf = dalloc<unsigned short>((size_t)B * Tp * A);
    d.encbf = dalloc<unsigned short>((size_t)B * Tp * Hd);
    d.Wsbf2 = dalloc<unsigned short>((size_t)D * A); d.encbf2 = dalloc<unsigned short>((size_t)B * Tp * Hd);
    d.dE = dalloc<float>((size_t)U * B * Tp); d.dHl = dalloc<float>((size_t)U * B * D); d.dH = dalloc<float>((size_t)B * D);
    d.dC = dalloc<float>((size_t)B * D); d.dXin0 = dalloc<float>((size_t)U * B * I0D); d.Q = dalloc<float>((size_t)U * B * A);
    d.dQ = dalloc<float>((size_t)U * B * A); d.duRows = dalloc<float>((size_t)B * A);
    d.rec[0] = d.dXin0; d.recLd[0] = I0D; d.recOff[0] = E + Hd;
    const size_t lds = bf_lds_bytes(d);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int IT = 400;
    for (int which = 0; which < 4; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            for (int it = 0; it < IT; ++it) {
                const int t = 1 + (it % (U - 2));
                if (which == 0) hipLaunchKernelGGL((dec_step_fwd_pf_kernel<LAS_CELL_LSTM, 10>), dim3(B), dim3(RNT), lds, 0, d, t);
                if (which == 1) hipLaunchKernelGGL((dec_step_bwd_pf_kernel<LAS_CELL_LSTM, 10>), dim3(B), dim3(RNT), lds, 0, d, t, t - 1);
                if (which == 2) hipLaunchKernelGGL((dec_step_fwd_bf_kernel<LAS_CELL_LSTM, 1>), dim3(B), dim3(RNT), lds, 0, d, t);
                if (which == 3) hipLaunchKernelGGL((dec_step_bwd_bf_kernel<LAS_CELL_LSTM, 1>), dim3(B), dim3(RNT), lds, 0, d, t, t - 1);
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s : %.2f us/launch  (%s)\n", which == 0 ? "fwd_pf" : which == 1 ? "bwd_pf" : which == 2 ? "fwd_bf" : "bwd_bf",
                            ms * 1e3f / IT, hipGetErrorString(hipGetLastError()));
        }
    }
#ifdef LAS_ROW_STAMPS
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((dec_step_fwd_pf_kernel<LAS_CELL_LSTM, 10>), dim3(B), dim3(RNT), lds, 0, d, 7 + i);
    hipDeviceSynchronize();
    unsigned long long hs[32];
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamps), sizeof(hs));
    for (int i = 1; i < 9; ++i) printf("stamp %d: +%.2f us\n", i, (double)(hs[i] - hs[0]) * 0.01);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((dec_step_bwd_pf_kernel<LAS_CELL_LSTM, 10>), dim3(B), dim3(RNT), lds, 0, d, 8 + i, 7 + i);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamps), sizeof(hs));
    for (int i = 11; i < 20; ++i) printf("bwd stamp %d: +%.2f us\n", i, (double)(hs[i] - hs[10]) * 0.01);
#endif
    return 0;
}
