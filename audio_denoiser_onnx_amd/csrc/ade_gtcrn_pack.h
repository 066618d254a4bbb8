// ade_gtcrn_pack.h — host-side packing of GTCRN-family weights (BN-folded tensor set) into the lane / tap layouts of the kernels in
// ade_kernels.hip.  Shared by the GTCRN engine (ade_engine.hip) and the H-GTCRN sub-engine (ade_hgtcrn.hip), whose network is the same
// blocks with a wider first convolution.  `Loader` is any type with `const float* get(name, {dims...})` and a status member `st`
// (ADE_OK until the first missing / mis-shaped tensor).
#pragma once

#include <cstring>
#include <string>
#include <vector>

#include "ade_internal.h"
#include "../../include/ade.h"

namespace ade {

// ---- weight arena builder: canonical kernel layouts, every tensor 64-byte aligned ------------------------
struct Arena {
    std::vector<float> f;
    size_t alloc(size_t n) {
        size_t off = (f.size() + 15) & ~(size_t)15;
        f.resize(off + n, 0.0f);
        return off;
    }
};

// PyTorch GRU rows of hidden unit j -> [3x8 ih | 3xH hh | 3 b_ih | 3 b_hh]
inline void pack_gru_lane(float* dst, const float* wih, const float* whh, const float* bih, const float* bhh, int H, int j) {
    for (int g = 0; g < 3; ++g)
        for (int k = 0; k < 8; ++k) dst[g * 8 + k] = wih[(g * H + j) * 8 + k];
    for (int g = 0; g < 3; ++g)
        for (int k = 0; k < H; ++k) dst[24 + g * H + k] = whh[(g * H + j) * H + k];
    for (int g = 0; g < 3; ++g) {
        dst[24 + 3 * H + g] = bih[g * H + j];
        dst[24 + 3 * H + 3 + g] = bhh[g * H + j];
    }
}

struct GtOff { size_t pw1, pw1_b, dw, dw_b, pw2, pw2_b, gru, fc, tra_rot; float s1, s2; };
struct DpOff { size_t intra_gru, inter_gru, inter_rot, fc[2], fc_b[2], ln_w[2], ln_b[2]; };

template <class Loader>
bool load_gt(Loader& L, Arena& A, const std::string& p, bool deconv, GtOff& o) {
    const float* pw1 = deconv ? L.get(p + "point_conv1.weight", {24, 16, 1, 1}) : L.get(p + "point_conv1.weight", {16, 24, 1, 1});
    const float* pw1b = L.get(p + "point_conv1.bias", {16});
    const float* a1 = L.get(p + "point_act.weight", {1});
    const float* dw = L.get(p + "depth_conv.weight", {16, 1, 3, 3});
    const float* dwb = L.get(p + "depth_conv.bias", {16});
    const float* a2 = L.get(p + "depth_act.weight", {1});
    const float* pw2 = deconv ? L.get(p + "point_conv2.weight", {16, 8, 1, 1}) : L.get(p + "point_conv2.weight", {8, 16, 1, 1});
    const float* pw2b = L.get(p + "point_conv2.bias", {8});
    const float* wih = L.get(p + "tra.att_gru.weight_ih_l0", {48, 8});
    const float* whh = L.get(p + "tra.att_gru.weight_hh_l0", {48, 16});
    const float* bih = L.get(p + "tra.att_gru.bias_ih_l0", {48});
    const float* bhh = L.get(p + "tra.att_gru.bias_hh_l0", {48});
    const float* fcw = L.get(p + "tra.att_fc.weight", {8, 16});
    const float* fcb = L.get(p + "tra.att_fc.bias", {8});
    if (L.st != ADE_OK) return false;
    o.pw1 = A.alloc(24 * 16);
    for (int ci = 0; ci < 24; ++ci)
        for (int co = 0; co < 16; ++co) A.f[o.pw1 + ci * 16 + co] = deconv ? pw1[ci * 16 + co] : pw1[co * 24 + ci];
    o.pw1_b = A.alloc(16);
    memcpy(&A.f[o.pw1_b], pw1b, 64);
    o.dw = A.alloc(9 * 16);
    for (int c = 0; c < 16; ++c)
        for (int kt = 0; kt < 3; ++kt)
            for (int kf = 0; kf < 3; ++kf) {
                // decoder ConvTranspose2d taps y[t,f] += W[kt][kf] h[t-kt*d, f+1-kf] == encoder-form taps flipped in kt and kf
                const int ekt = deconv ? 2 - kt : kt, ekf = deconv ? 2 - kf : kf;
                A.f[o.dw + (ekt * 3 + ekf) * 16 + c] = dw[c * 9 + kt * 3 + kf];
            }
    o.dw_b = A.alloc(16);
    memcpy(&A.f[o.dw_b], dwb, 64);
    o.pw2 = A.alloc(16 * 8);
    for (int ci = 0; ci < 16; ++ci)
        for (int co = 0; co < 8; ++co) A.f[o.pw2 + ci * 8 + co] = deconv ? pw2[ci * 8 + co] : pw2[co * 16 + ci];
    o.pw2_b = A.alloc(8);
    memcpy(&A.f[o.pw2_b], pw2b, 32);
    o.gru = A.alloc(16 * 78);
    for (int j = 0; j < 16; ++j) pack_gru_lane(&A.f[o.gru + j * 78], wih, whh, bih, bhh, 16, j);
    o.fc = A.alloc(8 * 17);
    for (int c = 0; c < 8; ++c) {
        for (int k = 0; k < 16; ++k) A.f[o.fc + c * 17 + k] = fcw[c * 16 + k];
        A.f[o.fc + c * 17 + 16] = fcb[c];
    }
    // k_tra gathers the 16 hidden values of a site with DPP row rotations: rotation s hands lane j the value of lane (j + dir * s) & 15
    {
        const int dir = dpp_row_ror_direction();
        if (dir == 0) { if (L.st == ADE_OK) L.st = ADE_ERR_DEVICE; return false; }
        o.tra_rot = A.alloc(16 * 64);
        for (int j = 0; j < 16; ++j)
            for (int sft = 0; sft < 16; ++sft) {
                const int k = (j + dir * sft) & 15;
                for (int g = 0; g < 3; ++g) A.f[o.tra_rot + j * 64 + g * 16 + sft] = whh[(g * 16 + j) * 16 + k];
                A.f[o.tra_rot + j * 64 + 48 + sft] = fcw[(j & 7) * 16 + k];
            }
    }
    o.s1 = a1[0];
    o.s2 = a2[0];
    return true;
}

template <class Loader>
bool load_dp(Loader& L, Arena& A, const std::string& p, DpOff& o) {
    o.intra_gru = A.alloc(16 * 42);
    o.inter_gru = A.alloc(16 * 54);
    o.inter_rot = A.alloc(16 * 24);
    const int dir = dpp_row_ror_direction();
    if (dir == 0) { if (L.st == ADE_OK) L.st = ADE_ERR_DEVICE; return false; }
    for (int grp = 0; grp < 2; ++grp) {
        const std::string r = p + "intra_rnn.rnn" + std::to_string(grp + 1) + ".";
        for (int dir = 0; dir < 2; ++dir) {
            const std::string sfx = dir ? "_reverse" : "";
            const float* wih = L.get(r + "weight_ih_l0" + sfx, {12, 8});
            const float* whh = L.get(r + "weight_hh_l0" + sfx, {12, 4});
            const float* bih = L.get(r + "bias_ih_l0" + sfx, {12});
            const float* bhh = L.get(r + "bias_hh_l0" + sfx, {12});
            if (L.st != ADE_OK) return false;
            for (int j = 0; j < 4; ++j) pack_gru_lane(&A.f[o.intra_gru + (grp * 8 + dir * 4 + j) * 42], wih, whh, bih, bhh, 4, j);
        }
        const std::string q = p + "inter_rnn.rnn" + std::to_string(grp + 1) + ".";
        const float* wih = L.get(q + "weight_ih_l0", {24, 8});
        const float* whh = L.get(q + "weight_hh_l0", {24, 8});
        const float* bih = L.get(q + "bias_ih_l0", {24});
        const float* bhh = L.get(q + "bias_hh_l0", {24});
        if (L.st != ADE_OK) return false;
        for (int j = 0; j < 8; ++j) pack_gru_lane(&A.f[o.inter_gru + (grp * 8 + j) * 54], wih, whh, bih, bhh, 8, j);
        // k_inter_gru's DPP form: lane = 2 * unit + group, rotation by 2 s hands it the hidden value of unit (unit + dir * s) & 7 of its own group
        for (int unit = 0; unit < 8; ++unit)
            for (int g = 0; g < 3; ++g)
                for (int sft = 0; sft < 8; ++sft)
                    A.f[o.inter_rot + (2 * unit + grp) * 24 + g * 8 + sft] = whh[(g * 8 + unit) * 8 + ((unit + dir * sft) & 7)];
    }
    const char* part[2] = {"intra", "inter"};
    for (int i = 0; i < 2; ++i) {
        const float* fw = L.get(p + part[i] + "_fc.weight", {16, 16});
        const float* fb = L.get(p + part[i] + "_fc.bias", {16});
        const float* lw = L.get(p + part[i] + "_ln.weight", {kFw, 16});
        const float* lb = L.get(p + part[i] + "_ln.bias", {kFw, 16});
        if (L.st != ADE_OK) return false;
        o.fc[i] = A.alloc(256);
        for (int k = 0; k < 16; ++k)
            for (int co = 0; co < 16; ++co) A.f[o.fc[i] + k * 16 + co] = fw[co * 16 + k];
        o.fc_b[i] = A.alloc(16);
        memcpy(&A.f[o.fc_b[i]], fb, 64);
        o.ln_w[i] = A.alloc(kFw * 16);
        memcpy(&A.f[o.ln_w[i]], lw, kFw * 64);
        o.ln_b[i] = A.alloc(kFw * 16);
        memcpy(&A.f[o.ln_b[i]], lb, kFw * 64);
    }
    return true;
}

// banded form of a dense (n_in x n_out) row-major matrix: per output column the run [first nz, last nz]
inline void band_table(const float* m, int n_in, int n_out, std::vector<int>& start, std::vector<float>& w, int& count) {
    start.assign(n_out, 0);
    std::vector<int> len(n_out, 0);
    count = 0;
    for (int o = 0; o < n_out; ++o) {
        int lo = -1, hi = -1;
        for (int i = 0; i < n_in; ++i)
            if (m[(size_t)i * n_out + o] != 0.0f) { if (lo < 0) lo = i; hi = i; }
        if (lo >= 0) { start[o] = lo; len[o] = hi - lo + 1; }
        if (len[o] > count) count = len[o];
    }
    if (count == 0) count = 1;
    w.assign((size_t)count * n_out, 0.0f);
    for (int o = 0; o < n_out; ++o)
        for (int n = 0; n < len[o]; ++n) w[(size_t)n * n_out + o] = m[(size_t)(start[o] + n) * n_out + o];
}

}  // namespace ade
