// Unit check of the bf16 attention core of csrc/ade_melband.hip (TEST INFRASTRUCTURE): k_attention16 against a double-precision host softmax(q k^T) v on the same
// bf16 inputs.  Built two ways by tests/test_gemm16.py: g++ + tests/hipsim (CPU) and hipcc --offload-arch=gfx950 (GPU).     usage: melband16_unit [n heads nseq stride_mode ...]
#include "../../audio_denoiser_onnx_amd/csrc/ade_melband.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace ade {
namespace {

unsigned short h_bf16(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
float h_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// stride_mode 0: a sequence = n consecutive rows (time axis); 1: rows nseq apart (frequency axis)
int attention_case(int n, int heads, int nseq, int stride_mode) {
    const int di = heads * kDh, ldq = 3 * di + 8 * ((heads + 7) / 8), R = n * nseq;
    const long long seq_stride = stride_mode ? 1 : n, pos_stride = stride_mode ? nseq : 1;
    std::vector<unsigned short> q((size_t)R * ldq);
    unsigned s = 777u + n * 13 + heads;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : q) v = h_bf16(rnd() * 1.5f);
    unsigned short *dq, *dao;
    (void)hipMalloc((void**)&dq, q.size() * 2);
    (void)hipMalloc((void**)&dao, (size_t)R * di * 2);
    (void)hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(dao, 0, (size_t)R * di * 2);
    if (n > 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attention16<2>), dim3((unsigned)nseq, (unsigned)heads, (unsigned)((n + 127) / 128)), dim3(256), 0, (hipStream_t)0,
                                   (const gemm16::bf16_t*)dq, dao, n, seq_stride, pos_stride, ldq, di);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attention16<1>), dim3((unsigned)nseq, (unsigned)heads, 1), dim3(256), 0, (hipStream_t)0, (const gemm16::bf16_t*)dq, dao, n,
                            seq_stride, pos_stride, ldq, di);
    (void)hipDeviceSynchronize();
    std::vector<unsigned short> ao((size_t)R * di);
    (void)hipMemcpy(ao.data(), dao, ao.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0.0;
    std::vector<double> p(n);
    for (int sq = 0; sq < nseq; ++sq)
        for (int hd = 0; hd < heads; ++hd)
            for (int i = 0; i < n; ++i) {
                const size_t qr = (size_t)(sq * seq_stride + i * pos_stride);
                double mx = -1e300;
                for (int j = 0; j < n; ++j) {
                    const size_t kr = (size_t)(sq * seq_stride + j * pos_stride);
                    double d = 0.0;
                    for (int c = 0; c < kDh; ++c) d += (double)h_f32(q[qr * ldq + hd * kDh + c]) * (double)h_f32(q[kr * ldq + di + hd * kDh + c]);
                    p[j] = d;
                    mx = d > mx ? d : mx;
                }
                double sum = 0.0;
                for (int j = 0; j < n; ++j) { p[j] = exp(p[j] - mx); sum += p[j]; }
                const double gate = 1.0 / (1.0 + exp(-(double)h_f32(q[qr * ldq + 3 * di + hd])));
                for (int c = 0; c < kDh; ++c) {
                    double o = 0.0;
                    for (int j = 0; j < n; ++j) o += p[j] * (double)h_f32(q[(size_t)(sq * seq_stride + j * pos_stride) * ldq + 2 * di + hd * kDh + c]);
                    o = o / sum * gate;
                    const double got = (double)h_f32(ao[qr * di + hd * kDh + c]);
                    worst = fmax(worst, fabs(got - o));
                }
            }
    const bool ok = worst < 0.02;
    printf("attention16 n=%d heads=%d nseq=%d mode=%d: max|d| %.3e -> %s\n", n, heads, nseq, stride_mode, worst, ok ? "OK" : "FAIL");
    (void)hipFree(dq); (void)hipFree(dao);
    return ok ? 0 : 1;
}

}  // namespace
}  // namespace ade

int main(int argc, char** argv) {
    int rc = 0;
    if (argc < 5) { rc |= ade::attention_case(70, 1, 2, 0); rc |= ade::attention_case(40, 2, 3, 1); return rc; }
    for (int i = 1; i + 3 < argc; i += 4) rc |= ade::attention_case(atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3]));
    return rc;
}
