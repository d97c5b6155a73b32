// wire.hpp — wire / disk formats around the path (SRS files, ptau, commitments, proof header).
// Part of zolt_host.hpp (the C++ host mirror over include/zolt_gpu.h); included by it, after the parts it depends on.
#pragma once
#ifndef ZOLT_HOST_UMBRELLA
#error "include zolt_host.hpp"
#endif
namespace zolt {

// ---------------------------------------------------------------- wire / disk formats around the path
// G1 coordinates travel as big-endian canonical bytes (commitments, raw SRS) or little-endian canonical bytes (ptau); the conversion
// to Montgomery limbs and the curve check run on the device (zg_field_op, zg_g1_is_on_curve_batch).
namespace wire {
struct SRSError : std::runtime_error { using std::runtime_error::runtime_error; };  // TruncatedData, InvalidFileFormat, UnsupportedFormat, PointNotOnCurve
struct G1Points {
    std::vector<uint64_t> xy;  // n * 8 Montgomery limbs (x | y), zeros at infinity
    std::vector<uint8_t> inf;
    size_t size() const { return inf.size(); }
};
inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
inline uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | (uint64_t)le32(p + 4) << 32; }
// `count` records of 64 bytes (x | y), big- or little-endian integers; all-zero = infinity; every other point checked on the curve
// (parseG1Uncompressed, src/poly/commitment/srs.zig:65-99; parseG1LE, :616-660)
inline G1Points g1FromBytes(const uint8_t *rec, size_t count, bool big_endian) {
    G1Points out;
    out.xy.assign(count * 8, 0);
    out.inf.assign(count, 0);
    if (!count) return out;
    std::vector<uint64_t> raw(count * 8);
    for (size_t i = 0; i < count; i++) {
        bool any = false;
        for (size_t b = 0; b < 64; b++) any = any || rec[64 * i + b] != 0;
        out.inf[i] = any ? 0 : 1;
        for (size_t c = 0; c < 2; c++)
            for (size_t l = 0; l < 4; l++) {
                uint64_t v = 0;
                for (size_t b = 0; b < 8; b++) {
                    const size_t byte_le = 8 * l + b;  // byte index counted from the least significant end
                    v |= (uint64_t)rec[64 * i + 32 * c + (big_endian ? 31 - byte_le : byte_le)] << (8 * b);
                }
                raw[8 * i + 4 * c + l] = v;
            }
    }
    check(zg_field_op(ZG_FIELD_FP, ZG_OP_TO_MONT, raw.data(), nullptr, out.xy.data(), 2 * count), "zg_field_op");  // reduces like Fp.fromBytes
    for (size_t i = 0; i < count; i++)
        if (out.inf[i]) std::fill(out.xy.begin() + 8 * i, out.xy.begin() + 8 * i + 8, 0);
    std::vector<uint8_t> ok(count);
    check(zg_g1_is_on_curve_batch(out.xy.data(), out.inf.data(), count, ok.data()), "zg_g1_is_on_curve_batch");
    for (uint8_t v : ok)
        if (!v) throw SRSError("PointNotOnCurve");
    return out;
}
// G1 part of loadFromRawBinary (srs.zig:256-306): u32 n (LE) | n x (x BE | y BE) | 128 B tau G2 | 64 B G1 | 128 B G2 (the trailer comes back raw)
inline G1Points srsG1FromRaw(const std::vector<uint8_t> &data, std::vector<uint8_t> *trailer = nullptr) {
    if (data.size() < 4) throw SRSError("TruncatedData");
    const size_t n = le32(data.data());
    if (data.size() < 4 + 64 * n + 128 + 64 + 128) throw SRSError("TruncatedData");
    if (trailer) trailer->assign(data.begin() + 4 + 64 * n, data.end());
    return g1FromBytes(data.data() + 4, n, true);
}
// serializeToRawBinary's G1 section (srs.zig:358-408): toBytesBE of x and y; identity = 64 zero bytes
inline std::vector<uint8_t> g1ToBytesBE(const G1Points &pts) {
    const size_t n = pts.size();
    std::vector<uint64_t> canon(n * 8);
    if (n) check(zg_field_op(ZG_FIELD_FP, ZG_OP_FROM_MONT, pts.xy.data(), nullptr, canon.data(), 2 * n), "zg_field_op");
    std::vector<uint8_t> out(64 * n, 0);
    for (size_t i = 0; i < n; i++) {
        if (pts.inf[i]) continue;
        for (size_t c = 0; c < 2; c++)
            for (size_t l = 0; l < 4; l++)
                for (size_t b = 0; b < 8; b++) out[64 * i + 32 * c + 31 - (8 * l + b)] = (uint8_t)(canon[8 * i + 4 * c + l] >> (8 * b));
    }
    return out;
}
inline std::vector<uint8_t> srsG1ToRaw(const G1Points &pts, const std::vector<uint8_t> &trailer = std::vector<uint8_t>(128 + 64 + 128, 0)) {
    std::vector<uint8_t> out(4);
    for (int b = 0; b < 4; b++) out[b] = (uint8_t)(pts.size() >> (8 * b));
    auto body = g1ToBytesBE(pts);
    out.insert(out.end(), body.begin(), body.end());
    out.insert(out.end(), trailer.begin(), trailer.end());
    return out;
}
// PolyCommitment.toBytes / fromBytes (src/zkvm/commitment_types.zig:49-65): x || y big-endian canonical, identity = 64 zero bytes
inline std::array<uint8_t, 64> commitmentToBytes(const AffinePoint &p) {
    G1Points one;
    one.xy.assign(8, 0);
    one.inf.assign(1, p.infinity ? 1 : 0);
    if (!p.infinity) {
        std::memcpy(one.xy.data(), p.x.limbs, 32);
        std::memcpy(one.xy.data() + 4, p.y.limbs, 32);
    }
    auto v = g1ToBytesBE(one);
    std::array<uint8_t, 64> out;
    std::copy(v.begin(), v.end(), out.begin());
    return out;
}
// G1 side of loadFromPtau (srs.zig:733-900, snarkjs powers-of-tau container): "ptau" | u32 version (= 1) | u32 sections | sections (u32 type,
// u64 size, payload); header payload: u32 field size (= 32) | 32-byte prime | u32 power | u32 ceremony power
struct Ptau {
    uint32_t power = 0, ceremony_power = 0;
    G1Points powers_of_tau_g1, alpha_tau_g1, beta_tau_g1;
    bool has_alpha = false, has_beta = false;
    std::vector<uint8_t> tau_g2_raw, beta_g2_raw;  // pairing side: out of scope, returned untouched
};
inline Ptau srsG1FromPtau(const std::vector<uint8_t> &data) {
    if (data.size() < 12) throw SRSError("TruncatedData");
    if (std::memcmp(data.data(), "ptau", 4) != 0) throw SRSError("InvalidFileFormat");
    if (le32(data.data() + 4) != 1) throw SRSError("UnsupportedFormat");
    const uint32_t nsec = le32(data.data() + 8);
    size_t off = 12;
    std::map<uint32_t, std::pair<size_t, size_t>> secs;  // type -> (offset, size); a later section of the same type wins
    for (uint32_t i = 0; i < nsec; i++) {
        if (off + 12 > data.size()) throw SRSError("TruncatedData");
        const uint32_t typ = le32(data.data() + off);
        const uint64_t size = le64(data.data() + off + 4);
        off += 12;
        if (size > data.size() - off) throw SRSError("TruncatedData");
        secs[typ] = {off, (size_t)size};
        off += (size_t)size;
    }
    if (!secs.count(1)) throw SRSError("InvalidFileFormat");
    const auto hdr = secs[1];
    if (hdr.second < 8) throw SRSError("TruncatedData");
    if (le32(data.data() + hdr.first) != 32) throw SRSError("UnsupportedFormat");
    if (hdr.second < 44) throw SRSError("TruncatedData");
    Ptau out;
    out.power = le32(data.data() + hdr.first + 36);
    out.ceremony_power = le32(data.data() + hdr.first + 40);
    auto points = [&](uint32_t typ, size_t most) {
        const auto sec = secs[typ];
        return g1FromBytes(data.data() + sec.first, std::min(most, sec.second / 64), false);
    };
    if (secs.count(2)) out.powers_of_tau_g1 = points(2, (size_t(1) << out.power) * 2 - 1);
    if (secs.count(4)) { out.alpha_tau_g1 = points(4, size_t(1) << out.power); out.has_alpha = true; }
    if (secs.count(5)) { out.beta_tau_g1 = points(5, size_t(1) << out.power); out.has_beta = true; }
    if (secs.count(3)) out.tau_g2_raw.assign(data.begin() + secs[3].first, data.begin() + secs[3].first + secs[3].second);
    if (secs.count(6)) out.beta_g2_raw.assign(data.begin() + secs[6].first, data.begin() + secs[6].first + secs[6].second);
    return out;
}
// Header of serializeProof (src/zkvm/serialization.zig:283-306): "ZOLT" | u32 version 1 | bytecode proof {commitment, read_ts, write_ts,
// 32-byte legacy field element} | memory proof {commitment, final_state, read_ts, write_ts} | register proof {same four}: the eleven
// commitments this backend produces, in file order
static constexpr const char *PROOF_COMMITMENT_NAMES[11] = {
    "bytecode.commitment", "bytecode.read_ts_commitment", "bytecode.write_ts_commitment", "memory.commitment", "memory.final_state_commitment",
    "memory.read_ts_commitment", "memory.write_ts_commitment", "register.commitment", "register.final_state_commitment", "register.read_ts_commitment",
    "register.write_ts_commitment"};
inline std::array<std::array<uint8_t, 64>, 11> parseZoltProofCommitments(const std::vector<uint8_t> &data) {
    if (data.size() < 8 + 3 * 64 + 32 + 8 * 64 || std::memcmp(data.data(), "ZOLT", 4) != 0) throw std::invalid_argument("not a ZOLT proof");
    if (le32(data.data() + 4) != 1) throw std::invalid_argument("unsupported ZOLT proof version");
    std::array<std::array<uint8_t, 64>, 11> out;
    size_t off = 8;
    for (size_t i = 0; i < 11; i++) {
        if (i == 3) off += 32;  // bytecode._legacy_commitment
        std::memcpy(out[i].data(), data.data() + off, 64);
        off += 64;
    }
    return out;
}
inline std::vector<uint8_t> serializeZoltProofHeader(const std::array<std::array<uint8_t, 64>, 11> &commitments) {
    std::vector<uint8_t> out = {'Z', 'O', 'L', 'T', 1, 0, 0, 0};
    for (size_t i = 0; i < 11; i++) {
        if (i == 3) out.insert(out.end(), 32, 0);  // F.zero()
        out.insert(out.end(), commitments[i].begin(), commitments[i].end());
    }
    return out;  // the first 744 bytes of the proof
}
}  // namespace wire

}  // namespace zolt
