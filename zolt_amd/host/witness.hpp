// witness.hpp — the R1CS cycle inputs as INTEGER COLUMNS and the device-resident witness matrix built from them.
// Part of zolt_host.hpp (the C++ host mirror over include/zolt_gpu.h); included by it, after the parts it depends on.
//
// The reference widens every input of every cycle to a field element on the CPU (R1CSWitnessGenerator.generateWitness,
// src/zkvm/r1cs/constraints.zig:1469-1494: createNoopWitness :1418-1438, fromTraceStep :929-1223) and the stage provers read that
// 43-column matrix. Every one of those values is F.fromU64 of a machine word, signedI64ToField of an immediate (:868-876), a 0/1 flag,
// or a sum / product of two such values, so the row also exists as 156 bytes of integers; zg_fr_rows_from_columns widens them in HBM to
// the identical 1376-byte row. CycleColumns::fromTrace is the integer-domain restatement of the generator (what a Zig shim computes in
// place of the field-element rows); CycleWitnessMatrix is the ONE resident matrix the stages share.
#pragma once
#ifndef ZOLT_HOST_UMBRELLA
#error "include zolt_host.hpp"
#endif
namespace zolt {

// tracer.TraceStep as fromTraceStep reads it (src/tracer/mod.zig:14-45)
struct R1CSTraceStep {
    uint32_t instruction = 0;
    uint64_t pc = 0, unexpanded_pc = 0, rs1_value = 0, rs2_value = 0, rd_value = 0;
    bool has_memory_value = false;
    uint64_t memory_value = 0;
    bool is_compressed = false, is_noop = false;
};

struct CycleColumns {
    static constexpr size_t NUM_INPUTS = 43;
    // R1CSInputIndex (:38-86) of the columns by storage class
    static constexpr int U64_INPUTS[12] = {0, 6, 7, 10, 11, 12, 13, 14, 15, 17, 18, 21};  // Left, PC, UnexpandedPC, Rs1, Rs2, RdWrite, RamRead, RamWrite, LeftLookup, NextUnexpandedPC, NextPC, LookupOutput
    static constexpr int WIDE_INPUTS[3] = {1, 9, 16};                                      // RightInstructionInput, RamAddress, RightLookupOperand
    static constexpr int BIT_INPUTS[24] = {3, 4, 5, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42};
    enum Bit : uint32_t { WriteLookupOutputToRD, WritePCtoRD, ShouldBranch, ShouldJump, AddOperands, SubtractOperands, MultiplyOperands, Load, Store, Jump,
                          WriteLookupOutputToRDFlag, VirtualInstruction, Assert, DoNotUpdateUnexpandedPC, Advice, IsCompressed, IsFirstInSequence, IsRdNotZero,
                          Branch, IsNoop, LeftOperandIsRs1, LeftOperandIsPC, RightOperandIsRs2, RightOperandIsImm };

    // The columns live in ONE slab of pinned host memory (zg_host_alloc): the copies to the device then run by DMA at link rate and do not
    // depend on the page state of the process (BENCH_r04: the same pageable upload took 26 ms on one box and 55 ms on another). The slab is
    // kept per thread and reused by the next trace (hipHostMalloc of 160 MB costs more than the upload it speeds up).
    size_t n = 0;
    uint64_t *u64[12] = {};
    int64_t *imm = nullptr;
    uint64_t *wide[3] = {};    // 2 words per row: 128-bit two's complement
    uint32_t *word = nullptr;  // the 24 single-bit inputs of a cycle

    explicit CycleColumns(size_t cycles = 0, bool zeroed = true) { resize(cycles, zeroed); }
    CycleColumns(const CycleColumns &) = delete;
    CycleColumns &operator=(const CycleColumns &) = delete;
    CycleColumns(CycleColumns &&o) noexcept { *this = std::move(o); }
    CycleColumns &operator=(CycleColumns &&o) noexcept {
        release();
        std::memcpy(static_cast<void *>(this), &o, sizeof(*this));
        o.slab_ = nullptr;
        o.n = 0;
        return *this;
    }
    ~CycleColumns() { release(); }
    static constexpr size_t BYTES_PER_CYCLE = 12 * 8 + 8 + 3 * 16 + 4;  // 156
    // bytes of the columns of `rows` cycles laid out back to back, every column on a 256-byte boundary
    static size_t regionBytes(size_t rows) { return (((rows + 31) & ~size_t(31)) * BYTES_PER_CYCLE + 255) & ~size_t(255); }
    void resize(size_t cycles, bool zeroed = true) {
        if (cycles == 0) {  // nothing to hold: no slab
            release();
            layout(nullptr, 0);
            return;
        }
        allocate(regionBytes(cycles) + 256);
        if (zeroed) std::memset(slab_, 0, bytes_);
        layout(static_cast<char *>(slab_), cycles);
    }
    // `owner` takes ONE slab for n cycles cut into slices of `per` cycles; the result holds one non-owning CycleColumns per slice, each
    // with its columns back to back in its own region of the slab — a slice then crosses PCIe as one contiguous copy
    // (zg_fr_rows_from_columns merges sources that sit side by side), while another slice is still being decoded
    static std::vector<CycleColumns> sliced(CycleColumns &owner, size_t n, size_t per) {
        size_t bytes = 256;
        for (size_t a = 0; a < n; a += per) bytes += regionBytes(std::min(per, n - a));
        owner.allocate(bytes);
        owner.layout(static_cast<char *>(owner.slab_), 0);
        std::vector<CycleColumns> part;
        char *p = static_cast<char *>(owner.slab_);
        for (size_t a = 0; a < n; a += per) {
            const size_t rows = std::min(per, n - a);
            part.emplace_back();  // (zero cycles: no slab of its own; the view owns nothing)
            part.back().layout(p, rows);
            p += regionBytes(rows);
        }
        return part;
    }
    static int64_t sx(uint64_t v, int bits) { return (int64_t)(v << (64 - bits)) >> (64 - bits); }
    static int64_t immOf(uint32_t w) {  // deriveImmediate (:1226-1274) as a signed integer
        const uint32_t op = w & 0x7F;
        if (op == 0x13 || op == 0x03 || op == 0x67) return sx(w >> 20, 12);
        if (op == 0x23) return sx((((w >> 25) & 0x7F) << 5) | ((w >> 7) & 0x1F), 12);
        if (op == 0x63) return sx((((w >> 31) & 1) << 12) | (((w >> 7) & 1) << 11) | (((w >> 25) & 0x3F) << 5) | (((w >> 8) & 0xF) << 1), 13);
        if (op == 0x6F) return sx((((w >> 31) & 1) << 20) | (((w >> 12) & 0xFF) << 12) | (((w >> 20) & 1) << 11) | (((w >> 21) & 0x3FF) << 1), 21);
        if (op == 0x37 || op == 0x17) return (int64_t)(w & 0xFFFFF000u);
        return 0;
    }
    static bool nextIsNoop(const R1CSTraceStep *s) {  // isNoopInstruction (:569-595)
        if (!s) return false;
        if (s->is_noop) return true;
        const uint32_t w = s->instruction;
        return (w & 0x7F) == 0x13 && ((w >> 7) & 31) == 0 && ((w >> 15) & 31) == 0 && ((w >> 12) & 7) == 0 && (w >> 20) == 0;
    }
    void setWide(int k, size_t i, __int128 v) {
        wide[k][2 * i] = (uint64_t)(unsigned __int128)v;
        wide[k][2 * i + 1] = (uint64_t)((unsigned __int128)v >> 64);
    }

    // the integer-domain restatement of generateWitness over a NoOp-padded trace
    static CycleColumns fromTrace(const std::vector<R1CSTraceStep> &steps) {
        CycleColumns c(steps.size(), false);  // every decoding thread clears its own rows (164 MB of memset on one thread cost more than the decode)
        c.decodeParallel(steps, 0, c.n);
        return c;
    }
    // rows [i0, i1) of the columns from steps [step0 + i0, step0 + i1) of the trace (a row reads its step and the step after it, nothing else)
    void decodeRange(const std::vector<R1CSTraceStep> &steps, size_t i0, size_t i1, size_t step0 = 0) {
        CycleColumns &c = *this;
        uint64_t *Left = c.u64[0], *PC = c.u64[1], *UPC = c.u64[2], *Rs1 = c.u64[3], *Rs2 = c.u64[4], *RdW = c.u64[5], *RamR = c.u64[6], *RamW = c.u64[7],
                 *LeftLookup = c.u64[8], *NextUPC = c.u64[9], *NextPC = c.u64[10], *Lookup = c.u64[11];
        for (auto *col : c.u64) std::memset(col + i0, 0, (i1 - i0) * 8);
        for (auto *col : c.wide) std::memset(col + 2 * i0, 0, (i1 - i0) * 16);
        std::memset(c.imm + i0, 0, (i1 - i0) * 8);
        for (size_t i = i0; i < i1; i++) {
            const R1CSTraceStep &st = steps[step0 + i];
            uint32_t bits = 0;
            auto set = [&](Bit b) { bits |= 1u << b; };
            if (st.is_noop) {  // createNoopWitness (:1418-1438)
                set(DoNotUpdateUnexpandedPC);
                set(IsNoop);
                c.word[i] = bits;
                continue;
            }
            const R1CSTraceStep *nx = step0 + i + 1 < steps.size() ? &steps[step0 + i + 1] : nullptr;
            const uint32_t w = st.instruction, op = w & 0x7F, f3 = (w >> 12) & 7, f7 = (w >> 25) & 0x7F, rd = (w >> 7) & 31;
            const bool load = op == 0x03, store = op == 0x23, branch = op == 0x63;
            if (load) set(Load);
            if (store) set(Store);
            if (st.is_compressed) set(IsCompressed);
            const int64_t im = immOf(w);
            c.imm[i] = im;
            const bool reads1 = op == 0x13 || op == 0x03 || op == 0x67 || op == 0x1B || op == 0x33 || op == 0x3B || op == 0x23 || op == 0x63;  // :957-977
            const bool reads2 = op == 0x33 || op == 0x3B || op == 0x23 || op == 0x63;                                                           // :986-993
            const uint64_t rs1 = reads1 ? st.rs1_value : 0, rs2 = reads2 ? st.rs2_value : 0;
            Rs1[i] = rs1;
            Rs2[i] = rs2;
            if (load || store) c.setWide(1, i, (__int128)st.rs1_value + im);  // RamAddress = rs1 + imm in the field (:1001-1009)
            const uint64_t mem = st.has_memory_value ? st.memory_value : 0;
            if (load) RamR[i] = RamW[i] = RdW[i] = mem;                        // :1023-1047
            else if (store) { RamR[i] = mem; RamW[i] = st.rs2_value; }
            else if (!branch && rd != 0) RdW[i] = st.rd_value;
            const bool l_rs1 = op == 0x33 || op == 0x13 || op == 0x03 || op == 0x67 || op == 0x23 || op == 0x63 || op == 0x1B || op == 0x3B;  // :1059-1098
            const bool l_pc = op == 0x17 || op == 0x6F, r_rs2 = op == 0x33 || op == 0x63 || op == 0x3B;
            const bool r_imm = op == 0x13 || op == 0x03 || op == 0x67 || op == 0x23 || op == 0x37 || op == 0x17 || op == 0x6F || op == 0x1B;
            if (l_rs1) set(LeftOperandIsRs1);
            if (l_pc) set(LeftOperandIsPC);
            if (r_rs2) set(RightOperandIsRs2);
            if (r_imm) set(RightOperandIsImm);
            const uint64_t left = l_rs1 ? rs1 : (l_pc ? st.unexpanded_pc : 0);                                  // :1106-1118
            const __int128 right = r_rs2 ? (__int128)rs2 : (r_imm ? (__int128)im : (__int128)0);
            Left[i] = left;
            c.setWide(0, i, right);
            uint64_t lookup;  // computeLookupOutput (:600-640)
            if (op == 0x6F) lookup = st.pc + (uint64_t)im;
            else if (op == 0x67) lookup = (st.rs1_value + (uint64_t)sx(w >> 20, 12)) & ~uint64_t(1);
            else if (branch) {
                const uint64_t a = st.rs1_value, b = st.rs2_value;
                const int64_t sa = (int64_t)a, sb = (int64_t)b;
                lookup = f3 == 0 ? a == b : f3 == 1 ? a != b : f3 == 4 ? sa < sb : f3 == 5 ? sa >= sb : f3 == 6 ? a < b : f3 == 7 ? a >= b : 0;
            } else lookup = st.rd_value;
            Lookup[i] = lookup;
            PC[i] = st.pc;
            UPC[i] = st.unexpanded_pc;
            if (nx && !nx->is_noop) { NextPC[i] = nx->pc; NextUPC[i] = nx->unexpanded_pc; }  // :1150-1172
            // setFlagsFromInstruction (:1288-1398): circuit flags and the two lookup operands
            // RightLookupOperand: the rows of a MUL take Product (constraint 9) — a full-width product fits no signed 128-bit word, so the
            // device adds Product * FlagMultiplyOperands to this column (descriptors()) — every other row its sum / difference / pass-through
            bool wl = false, jump = false, zero_left = false;
            __int128 lookup_right = right;
            if (op == 0x33) {
                if (f7 == 0x01) {
                    if (f3 == 0) {
                        set(MultiplyOperands);
                        zero_left = true;
                        lookup_right = 0;
                    }
                } else if (f7 == 0x20 && f3 == 0) {
                    set(SubtractOperands);
                    zero_left = true;
                    lookup_right = (__int128)left - right + ((__int128)1 << 64);
                } else {
                    set(AddOperands);
                    zero_left = true;
                    lookup_right = (__int128)left + right;
                }
                wl = true;
            } else if (op == 0x13 || op == 0x37 || op == 0x17) {
                set(AddOperands);
                zero_left = true;
                lookup_right = (__int128)left + right;
                wl = true;
            } else if (op == 0x6F || op == 0x67) {
                set(AddOperands);
                zero_left = true;
                lookup_right = (__int128)left + right;
                jump = true;
            }
            LeftLookup[i] = zero_left ? 0 : left;
            c.setWide(2, i, lookup_right);
            if (wl) set(WriteLookupOutputToRDFlag);
            if (jump) {
                set(Jump);
                if (!nextIsNoop(nx)) set(ShouldJump);  // :1181-1185
            }
            if (rd != 0) {  // :1190-1215
                set(IsRdNotZero);
                if (wl) set(WriteLookupOutputToRD);
                if (jump) set(WritePCtoRD);
            }
            if (branch) {
                set(Branch);
                if (lookup) set(ShouldBranch);
            }
            c.word[i] = bits;
        }
    }
    // every cycle is independent: long ranges are decoded by several host threads (ZOLT_HOST_THREADS, default up to 16 — the loop is
    // bound by its 17 output streams, 8 -> 16 threads took 4.7 -> 2.5 ms at 2^20 cycles, more gained nothing)
    static size_t hostThreads(size_t rows) {
        size_t t = rows >= (size_t(1) << 16) ? std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 16) : 1;
        if (const char *e = std::getenv("ZOLT_HOST_THREADS")) t = std::max(1, atoi(e));
        return t;
    }
    void decodeParallel(const std::vector<R1CSTraceStep> &steps, size_t i0, size_t i1) {
        const size_t rows = i1 - i0, nthreads = hostThreads(rows);
        if (nthreads <= 1) {
            decodeRange(steps, i0, i1);
            return;
        }
        std::vector<std::thread> pool;
        const size_t per = (rows + nthreads - 1) / nthreads;
        for (size_t t = 0; t < nthreads; t++) {
            const size_t a = i0 + t * per, b = std::min(i1, a + per);
            if (a < b) pool.emplace_back([this, &steps, a, b] { decodeRange(steps, a, b); });
        }
        for (auto &th : pool) th.join();
    }

    // the 43 descriptors of zg_fr_rows_from_columns, in R1CSInputIndex order
    std::vector<zg_col_t> descriptors() const {
        std::vector<zg_col_t> d(NUM_INPUTS, zg_col_t{ZG_COL_ZERO, 0, 0, nullptr, nullptr});  // NextIsVirtual, NextIsFirstInSequence stay zero (:1160-1171)
        for (int k = 0; k < 12; k++) d[U64_INPUTS[k]] = zg_col_t{ZG_COL_U64, 0, 0, u64[k], nullptr};
        d[8] = zg_col_t{ZG_COL_I64, 0, 0, imm, nullptr};
        for (int k = 0; k < 2; k++) d[WIDE_INPUTS[k]] = zg_col_t{ZG_COL_I128, 0, 0, wide[k], nullptr};
        d[2] = zg_col_t{ZG_COL_MUL, 0, 1, nullptr, nullptr};  // Product = LeftInstructionInput * RightInstructionInput (:1120-1122)
        d[16] = zg_col_t{ZG_COL_MUL, 2, 25, wide[2], nullptr};  // RightLookupOperand = Product * FlagMultiplyOperands + the other rows' value
        for (uint32_t b = 0; b < 24; b++) d[BIT_INPUTS[b]] = zg_col_t{ZG_COL_BIT, b, 4, word, nullptr};
        return d;
    }
    size_t bytesPerCycle() const { return BYTES_PER_CYCLE; }

private:
    struct Slab { void *p = nullptr; size_t bytes = 0; ~Slab() { if (p) zg_host_free(p); } };
    static Slab &spare() { static thread_local Slab s; return s; }
    void release() {  // the slab goes back to the thread's spare slot (one is kept; a second one is freed)
        if (!slab_) return;
        Slab &keep = spare();
        if (!keep.p) { keep.p = slab_; keep.bytes = cap_; }
        else zg_host_free(slab_);
        slab_ = nullptr;
    }
    void allocate(size_t bytes) {  // the thread's spare slab when it is large enough, a new pinned one otherwise
        release();
        bytes_ = bytes;
        Slab &keep = spare();
        if (keep.p && keep.bytes >= bytes_) {
            slab_ = keep.p;
            cap_ = keep.bytes;
            keep.p = nullptr;
        } else {
            if (keep.p) { zg_host_free(keep.p); keep.p = nullptr; }
            check(zg_host_alloc(bytes_, &slab_), "zg_host_alloc");
            cap_ = bytes_;
        }
    }
    void layout(char *p, size_t cycles) {  // the column pointers of `cycles` rows inside the region at p
        n = cycles;
        const size_t pad = (n + 31) & ~size_t(31);
        for (auto &c : u64) { c = reinterpret_cast<uint64_t *>(p); p += pad * 8; }
        imm = reinterpret_cast<int64_t *>(p); p += pad * 8;
        for (auto &c : wide) { c = reinterpret_cast<uint64_t *>(p); p += pad * 16; }
        word = reinterpret_cast<uint32_t *>(p);
    }
    void *slab_ = nullptr;
    size_t bytes_ = 0, cap_ = 0;
};

// The cycle-major witness matrix (num_cycles x 43 elements, src/zkvm/r1cs/evaluation.zig:55-122) resident in HBM: built once, shared by
// the stages that read it (StreamingOuterProver, R1CSInputEvaluator, the product-virtualisation first round, Stage3Prover).
class CycleWitnessMatrix {
public:
    size_t num_cycles = 0;
    static std::shared_ptr<CycleWitnessMatrix> fromColumns(const CycleColumns &c) {
        auto m = std::make_shared<CycleWitnessMatrix>();
        m->num_cycles = c.n;
        m->d_.alloc(c.n * CycleColumns::NUM_INPUTS * 32);
        const auto d = c.descriptors();
        check(zg_fr_rows_from_columns(d.data(), d.size(), c.n, m->d_.u64()), "zg_fr_rows_from_columns");
        return m;
    }
    // trace -> matrix. A long trace is cut into slices of cycles: while slice s crosses PCIe and is widened (this thread, inside
    // zg_fr_rows_from_columns), the host threads decode slice s + 1 — the decode (2.3 ms at 2^20 cycles on 16 threads) hides behind the
    // 164 MB of copies (3.0 ms) instead of preceding them; every slice has its columns side by side in the slab and crosses as one copy.
    // 2^20 cycles: 7.7 ms decode-then-upload -> 4.8 ms (profiles/r5i_*). ZOLT_WITNESS_SLICES overrides the count (1: one call).
    static std::shared_ptr<CycleWitnessMatrix> fromTrace(const std::vector<R1CSTraceStep> &steps) {
        const size_t n = steps.size();
        size_t slices = n >= (size_t(1) << 18) ? 8 : 1;
        if (const char *e = std::getenv("ZOLT_WITNESS_SLICES")) slices = std::max(1, atoi(e));
        const size_t per = ((n + slices - 1) / slices + 63) & ~size_t(63);  // slices start on 64-row boundaries: 512-byte aligned in every column
        if (slices <= 1 || per >= n) return fromColumns(CycleColumns::fromTrace(steps));
        CycleColumns slab(0, false);
        std::vector<CycleColumns> part = CycleColumns::sliced(slab, n, per);
        auto m = std::make_shared<CycleWitnessMatrix>();
        m->num_cycles = n;
        m->d_.alloc(n * CycleColumns::NUM_INPUTS * 32);
        // the decoding threads live for the whole call: thread t decodes its share of slice 0, of slice 1, ...; a slice is final when
        // every thread has counted itself off on it
        const size_t n_slices = (n + per - 1) / per, T = CycleColumns::hostThreads(n);
        std::mutex mu;
        std::condition_variable cv;
        std::vector<size_t> done(n_slices, 0);  // threads finished with slice s (under mu)
        bool failed = false;
        std::vector<std::thread> pool;
        struct Join { std::vector<std::thread> &p; ~Join() { for (auto &t : p) if (t.joinable()) t.join(); } } join{pool};  // also on an exception below: the threads write the slab and read steps
        for (size_t t = 0; t < T; t++)
            pool.emplace_back([&, t] {
                for (size_t sl = 0; sl < n_slices; sl++) {
                    const size_t a = sl * per, b = std::min(n, a + per), share = (b - a + T - 1) / T, i0 = std::min(b, a + t * share), i1 = std::min(b, i0 + share);
                    bool ok = true;
                    try { if (i0 < i1) part[sl].decodeRange(steps, i0 - a, i1 - a, a); } catch (...) { ok = false; }
                    bool last;
                    { std::lock_guard<std::mutex> lk(mu); if (!ok) failed = true; last = ++done[sl] == T || !ok; }
                    if (last) cv.notify_one();
                    if (!ok) return;
                }
            });
        for (size_t sl = 0; sl < n_slices; sl++) {
            const size_t a = sl * per, b = std::min(n, a + per);
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return done[sl] == T || failed; });
                if (failed) throw std::runtime_error("CycleWitnessMatrix::fromTrace: a decoding thread failed");
            }
            const auto d = part[sl].descriptors();
            check(zg_fr_rows_from_columns(d.data(), d.size(), b - a, m->d_.u64() + a * CycleColumns::NUM_INPUTS * 4), "zg_fr_rows_from_columns");
        }
        return m;
    }
    // ready rows of field elements (R1CSCycleInputs.values per cycle): the 1376-bytes-per-cycle upload of rounds 3 and 4
    static std::shared_ptr<CycleWitnessMatrix> fromWitnesses(const void *rows, size_t cycles) {
        auto m = std::make_shared<CycleWitnessMatrix>();
        m->num_cycles = cycles;
        m->d_.alloc(cycles * CycleColumns::NUM_INPUTS * 32);
        check(zg_memcpy_h2d(m->d_.p, rows, cycles * CycleColumns::NUM_INPUTS * 32), "zg_memcpy_h2d");
        return m;
    }
    const uint64_t *u64() const { return d_.u64(); }
    std::vector<Fr> toHost() const {
        std::vector<Fr> out(num_cycles * CycleColumns::NUM_INPUTS);
        if (!out.empty()) check(zg_memcpy_d2h(out.data(), d_.p, out.size() * 32), "zg_memcpy_d2h");
        return out;
    }

private:
    DeviceMem d_;
};

}  // namespace zolt
