// engine.hpp -- internal interfaces of the sealhip engine (context, device tables, kernel launchers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <array>
#include <atomic>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <cstdlib>
#include <vector>

#include "devmath.hpp"
#include "ntt_bounds.hpp"
#include "hostmath.hpp"

namespace sealhip
{
    constexpr int kMaxRows = 136;          // rows of one "polynomial" a launch can describe
    constexpr int kMaxModuli = 64;         // SEAL_COEFF_MOD_COUNT_MAX (util/defines.h:48)
    constexpr unsigned short kSkipRow = 0xFFFF;

    // Which prime each row of a polynomial uses (index into Engine::d_primes); kSkipRow = leave untouched.
    struct RowMap
    {
        int rows;
        unsigned short prime[kMaxRows];
    };

    // Optional gather source of the single-pass forward NTT: destination row r of destination polynomial P reads
    // base[b] + P*poly_stride[b] + src_row*N instead of its own storage (code = b<<15 | reduce<<14 | src_row;
    // kSkipRow = in place). `reduce` applies barrett_reduce_63 w.r.t. the destination prime on load, which is the
    // single-prime mod-up rule of multi_special_primes.cpp:99-108.
    struct NttSource
    {
        const u64 *base[2];
        std::size_t poly_stride[2];
        unsigned short code[kMaxRows];
        // launch-wide treatment of the loaded words (single-prime mod-up, multi_special_primes.cpp:99-108): 0 none,
        // 1 barrett_reduce_63 w.r.t. the row's prime, 2 one conditional subtraction (every source prime is below twice
        // every destination prime, so the canonical residue is x or x - p). Both are no-ops on rows whose source prime
        // does not exceed their own. 4: the CKKS mod-down with one special prime (multi_special_primes.cpp:262-273): the
        // word becomes -(x mod aux_p) as the integer aux_p - (x mod aux_p) (0 stays 0), aux_p = the special prime.
        // 5: the same, on a source row whose inverse transform left its top layer to the consumer (kNttDeferTop): the
        // transform reads the pairs (c, c + N/2) anyway and applies BackwardLazyLast (ntt.cpp:274-281) to them first;
        // aux_top = the special prime's {n^-1, its Shoup quotient, w n^-1, its Shoup quotient}.
        // 7: mode 5 plus the rest of the CKKS mod-down as the transform's store phase (multi_special_primes.cpp:291-302 and the
        // final add, evaluator.cpp:2363-2366): the transformed word t of row (polynomial pl, prime q) is not stored; instead
        // v = (prod[pl][q] + t) * P^-1 mod q goes into the ciphertext -- added to what is there, or, with c0_src set,
        // written as (c0_src + v, v) for components (0, 1). Floating-point instance at log n >= 15 only (ntt_can_fuse_moddown).
        int reduce_mode;
        u64 aux_p, aux_cr1;
        u64 aux_top[4];
        struct ModDownStore
        {
            const u64 *inv_p, *inv_p_shoup; // [k] P^-1 mod q_i and its Shoup quotient (KsDev::invP on the device)
            const u64 *prod;                // [2m polynomials][prod_stride]: rows q of the key-switch products
            std::size_t prod_stride;
            u64 *ct;                        // [m][ct_stride]: component c at c * k rows
            std::size_t ct_stride;
            const u64 *c0_src;              // optional, [m][c0_stride]
            std::size_t c0_stride;
            unsigned *tflags;               // optional transparency sink of the m ciphertexts (devmath.hpp note_nonzero)
        } md;
    };
    constexpr unsigned short kSrcReduce = 0x4000, kSrcSecond = 0x8000; // kSrcReduce: informational (rows that need it)

    // A/B knobs (SEALHIP_NTT_NO_TICKET, _GATHER_TICKET, _POLY_MAJOR, _POLY_GROUP, _WHOLE_ROW, _CANON_BARRETT,
    // SEALHIP_KS_MAC_GROUP, _KS_MODDOWN_UNFUSED, _KS_MODDOWN_STORE_UNFUSED, SEALHIP_LIFT_TOP_OFF) exist in the measurement-only
    // build (`make exp`, -DSEALHIP_NTT_EXPERIMENT) alone: the shipping library reads none of them, so no environment variable
    // can re-open a race or switch a proved path off. What the shipping library does read: the resource knobs
    // SEALHIP_WORKSPACE_MB / SEALHIP_HOST_CHUNK / SEALHIP_HOST_THREADS and the switches that run the REFERENCE'S OWN operation
    // sequences instead of the proved shortcuts: SEALHIP_NTT_EXACT_FWD, SEALHIP_NTT_EXACT_INV, SEALHIP_NTT_CANON_EXACT,
    // SEALHIP_NTT_NO_FP64.
    inline const char *exp_env(const char *name)
    {
#ifdef SEALHIP_NTT_EXPERIMENT
        return std::getenv(name);
#else
        (void)name;
        return nullptr;
#endif
    }

    constexpr int kNttCanonical = 1; // fuse the canonicalising wrapper (ntt.h:236-245 / :328-333)
    constexpr int kNttStrict = 2;    // Harvey-corrected forward butterflies
    constexpr int kNttAnyRep = 8;    // inverse: the consumer canonicalises, any representative below 2p may be stored
    constexpr int kNttReduceOut = 0x10; // forward, single-pass kernel: one more conditional subtraction, outputs in [0, 2p)
    constexpr int kNttApprox = 0x20;   // forward, with kNttAnyRep or a consumer that takes outputs below (2 + g) p: approximate quotient (ntt_bounds.hpp section 2)
    constexpr int kNttPolyMajor = 0x4000;        // forward half kernel: live positions grouped item by item (ntt.hip half_block_map; group size in bits 16-23)
    constexpr int kNttPolyMajorRequest = 0x8000; // (launcher-internal: resolved per arithmetic instance)
    constexpr int kNttSmallQuot = 0x1000000;     // (launcher-internal) canonical approximate-quotient launch: single-precision quotient estimate in the store
    constexpr int kNttXchgTop = 0x2000000;       // (launcher-internal, SEALHIP_NTT_XCHG_TOP builds) the two workgroups of a row share the top layer
    constexpr int kNttDebugNoSignal = 0x40; // forward half kernel: never send the hand-off signal (tests of the time-out path)
    // forward, single-pass kernel, with kNttReduceOut, in place: the producer of the rows has already applied the top layer
    // (gap N/2) -- bfv_lift2 does for the Bsk rows it writes. Each workgroup then loads its own half only: no second read of
    // the row, no duplicated products (both workgroups of a row compute the top layer's products otherwise), no hand-off.
    constexpr int kNttTopDone = 0x80;
    constexpr int kNttDeferTop = 4;  // inverse, single-pass kernels only: leave the top layer (gap N/2) to the consumer
    // Primes below this bound have double-precision twiddle tables: the single-pass kernels then run their butterflies
    // on the FP64 pipe (exact, devmath.hpp) whenever the launch promises nothing about representatives that only the
    // integer sequence would deliver (canonical outputs, or kNttAnyRep). SEALHIP_NTT_NO_FP64=1 switches it off.
    // (the bound is what the schedules' worst-case recurrences in ntt_bounds.hpp admit: 50 bits)
    constexpr u64 kFpPrimeBound = u64(1) << bounds::kFpPrimeBits;

    struct NttRound
    {
        int beta, wlo, whi;
    };
    struct NttPass
    {
        int logn, t, c, b_lo, flags, nrounds;
        NttRound rounds[4];
    };
    struct NttPlan
    {
        int logn, npass, flags;
        bool serial;
        NttPass pass[2];
    };
    NttPlan plan_ntt(int logn, bool inverse, int flags);

    // ---- device-resident constants of one BFV level (RNSTool, rns.cpp:539-729) ----
    struct RnsDev
    {
        int k, nB, B;               // |q|, |Bsk|, |B|
        u64 t;                      // plain modulus
        // q -> Bsk U {m_tilde}
        u64 q_mt_inv[kMaxModuli];   // m_tilde * (q^_i)^{-1} mod q_i   (fused multiply_poly_scalar + inv_punctured)
        u64 q_inv[kMaxModuli];      // (q^_i)^{-1} mod q_i
        const u64 *q_to_Bsk;        // [nB][k]
        u64 q_to_mt[kMaxModuli];    // [k] mod 2^32
        u64 inv_prod_q_mod_mt;      // mod 2^32
        // per Bsk prime
        u64 prod_q_mod_Bsk[kMaxModuli + 2], inv_prod_q_mod_Bsk[kMaxModuli + 2], inv_mt_mod_Bsk[kMaxModuli + 2];
        // B -> q, B -> m_sk
        u64 B_inv[kMaxModuli + 1];  // (B^_i)^{-1} mod B_i
        const u64 *B_to_q;          // [k][B]
        u64 B_to_msk[kMaxModuli + 1];
        u64 inv_prod_B_mod_msk;
        u64 prod_B_mod_q[kMaxModuli];
        u64 inv_q_last_mod_q[kMaxModuli];
        // constant-folded forms used by the fused BFV kernels (canonical results are unchanged):
        //   lift : out_j = ( sum_i t_i*lift_L1[j][i] + temp_j*lift_L2[j] ) mod b_j
        //   floor: tb_j  = ( in_j*floor_G1[j] + sum_i t_i*floor_G2[j][i] ) mod b_j,  t_i = in_i*floor_F0[i] mod q_i
        const u64 *lift_L1;  // [nB][k]  M_ji * m_tilde^{-1} mod b_j
        const u64 *floor_G2; // [nB][k]  -(M_ji * (prod q)^{-1} [* B^_j^{-1} for j < B]) mod b_j
        u64 lift_L2[kMaxModuli + 2];  // prod_q * m_tilde^{-1} mod b_j
        u64 floor_G1[kMaxModuli + 2]; // t * (prod q)^{-1} [* B^_j^{-1} for j < B] mod b_j
        u64 floor_F0[kMaxModuli];     // t * (q^_i)^{-1} mod q_i
        // the same constants with the deferred top inverse-NTT layer's factor folded in (index 0: lower half of a row,
        // n^{-1}; index 1: upper half, w * n^{-1}; ntt.cpp:393-402), so the fused floor kernel multiplies once, not twice
        // Index 2, 3: the same with 2^64 mod the row's prime on top -- the inverse NTT that forms the tensor product on load
        // leaves the Montgomery factor 2^-64 on every word (ntt.hip: DyadicSrc). One array indexed by 2 * mont + half: the
        // kernel selects its table with scalar arithmetic, not with a branch.
        u64 floor_F0_top[4][kMaxModuli], floor_F0_top_s[4][kMaxModuli];
        u64 floor_G1m_top[4][kMaxModuli + 2];
        // per-row prime constants next to the level constants (p, -p^-1 mod 2^64, floor(2^64 / p)): read at compile-time
        // offsets with scalar loads the compiler can batch -- going through PrimeDev[prime id] made every row loop wait for
        // two dependent scalar loads
        u64 q_p[kMaxModuli], q_ninv[kMaxModuli], q_rdp[kMaxModuli];
        u64 b_p[kMaxModuli + 2], b_ninv[kMaxModuli + 2], b_rdp[kMaxModuli + 2];
        u64 b_w1[kMaxModuli + 2], b_w1s[kMaxModuli + 2]; // top-layer forward twiddle of the Bsk rows and its Shoup quotient
        // Montgomery/Shoup companions of the folded constants (suffix m: times 2^64 mod the row's prime; s: Shoup)
        const u64 *lift_L1m, *floor_G2m, *B_to_qm; // [nB][k], [nB][k], [k][B]
        u64 lift_L2m[kMaxModuli + 2], floor_G1m[kMaxModuli + 2];
        u64 q_mt_inv_s[kMaxModuli], floor_F0_s[kMaxModuli];
        u64 B_to_mskm[kMaxModuli + 1];
        u64 inv_prod_B_mod_msk_s;
        u64 pBm[kMaxModuli], nBm[kMaxModuli]; // prod_B_mod_q * 2^64, (q - prod_B_mod_q) * 2^64  (mod q_i)
        // decrypt_scale_and_round (rns.cpp:1070-1126), constants folded where the results are canonical anyway:
        //   y_i = in_i * dsr_scale[i] mod q_i  (|gamma t|_qi times (q^_i)^{-1});  {t, gamma} part = sum_i y_i * dsr_to_t/g[i]
        u64 dsr_scale[kMaxModuli], dsr_scale_s[kMaxModuli], dsr_to_t[kMaxModuli], dsr_to_g[kMaxModuli];
        u64 dsr_neg_inv_q_t, dsr_neg_inv_q_g, dsr_inv_gamma_t, dsr_gamma;
        unsigned gamma_prime; // PrimeDev id of gamma
        int redc_small;                        // every REDC of the fused kernels provably lands below 2p
        // (32-bit: there are no sub-dword scalar loads; a 16-bit id became a VECTOR load with a vmcnt(0) wait in the middle
        //  of every row loop)
        unsigned q_prime[kMaxModuli];       // prime ids of q rows
        unsigned bsk_prime[kMaxModuli + 2]; // prime ids of Bsk rows (m_sk last)
    };

    // ---- device-resident constants of the hybrid key switch at level k (multi_special_primes.cpp) ----
    struct KsDev
    {
        int k, nsp, nd, n_all, n_total; // ct primes, special primes, digits at this level, first-level k, key primes
        int is_ckks, strict;
        unsigned row_prime[kMaxModuli]; // prime id of ext row r (r < k: r; r >= k: n_all + r - k); 32-bit: scalar loads
        // mod-up: for bundle j, element a (source row j*nsp+a): inv_punch and its shoup; punch[dst row][a]
        const u64 *modup; // layout: [nd][ (2*nsp) + rows*nsp ] see engine.cpp
        // mod-down
        u64 inv_hat[kMaxModuli], inv_hat_shoup[kMaxModuli]; // [nsp]  p^_j^{-1} mod p_j
        const u64 *neg_hat;                                  // [k][nsp]  -p^_j mod q_i
        u64 invP[kMaxModuli], invP_shoup[kMaxModuli];        // [k]  P^{-1} mod q_i
    };

    // ---- CKKSEncoder::decode constants for the first k primes (RNSBase of the level, rns.cpp:237-290; context.cpp:370-376)
    constexpr int kCkksMaxLimbs = 32;
    struct CkksDecodeDev
    {
        int k;
        u64 q[kCkksMaxLimbs];         // total_coeff_modulus, little-endian limbs
        u64 half[kCkksMaxLimbs];      // upper_half_threshold = (q + 1) >> 1
        u64 inv_punct[kCkksMaxLimbs]; // (q / q_i)^{-1} mod q_i
        u64 punct[kCkksMaxLimbs * kCkksMaxLimbs]; // q / q_i, limbs of row i at i * kCkksMaxLimbs
    };

    struct LevelTools
    {
        std::unique_ptr<HostRnsTool> host_rns; // BFV + CKKS (CKKS only uses inv_q_last_mod_q)
        RnsDev *d_rns = nullptr;
        KsDev *d_ks = nullptr;
        KsDev h_ks{};
        RnsDev h_rns{};
        std::vector<void *> owned; // device allocations to free
        RowMap map_q{}, map_bsk{}, map_key{}, map_qbsk{};
    };

    struct KSwitchKey
    {
        u64 *d_data = nullptr;
        std::uint32_t n_digits = 0;
        std::size_t words = 0;
    };

    // Optional per-launch timing with HIP events on the launch stream (bench.py's roofline line).
    struct ProfRecord
    {
        const char *tag;
        hipEvent_t start, stop;
        double units; // rows (NTT passes) or lanes processed by the launch
    };

    // Per-host-thread execution state of a context: the reference's Evaluator is re-entrant
    // (native/src/seal/evaluator.h:1375-1377: it holds only the context and an immutable map; temporaries come from the
    // MemoryPoolHandle of the call), so every host thread that calls into a context gets its own lane -- HIP stream,
    // temporaries arena, forward-NTT tickets, launch profiler, graph capture state -- and operations of different
    // threads overlap instead of serialising. Tables, level constants and keys are shared and immutable.
    struct HostStage; // pinned staging buffers and copy streams of the *_host batch entries (hostbatch.cpp)
    void free_host_stage(HostStage *s);
    struct Lane
    {
        int device = -1;
        HostStage *stage = nullptr;
        hipStream_t stream = nullptr;
        bool own_stream = false;
        unsigned long long alloc_generation = 0; // bumps when the arena or the ticket buffer is re-allocated
        bool capturing = false; // between sealhip_graph_capture_begin / _end: no allocation, no synchronisation
        bool prof_on = false;
        std::vector<ProfRecord> prof;
        unsigned *d_tickets = nullptr; // per-row tickets of the single-pass forward NTT
        std::size_t tickets_cap = 0;
        // Transparency sink (sealhip_transparency_sink): device flags, one word per ciphertext of the operation's batch.
        // tsink_cur is what the final kernel of the operation in flight writes to (null outside an Evaluator entry that
        // supports it: internal launches of composite operations must not touch the caller's flags)
        unsigned *tsink = nullptr, *tsink_cur = nullptr;
        std::size_t tsink_cap = 0;
        // ... and what the NEXT launch of a sink-capable final kernel writes to (set by pipeline.cpp's SinkArm around exactly
        // that launch: the chunk's first item of tsink_cur)
        unsigned *tsink_arm = nullptr;
        std::size_t tsink_base = 0; // first item of an enclosing chunk loop (op_apply_galois around op_switch_key)
        // Sticky failure flag of this lane's launches (today: the forward NTT's sibling hand-off timing out), one word of
        // host-mapped memory per lane: a time-out in one thread's launch must fail THAT thread's next host-visible point,
        // not be seen and cleared by another thread's
        unsigned *h_fault = nullptr, *d_fault = nullptr;
        void *ws = nullptr; // workspace arena (stream-ordered reuse)
        std::size_t ws_bytes = 0, ws_used = 0;
        std::size_t ws_floor = 0; // bytes at the front of the arena held by an enclosing operation
        std::size_t ws_budget = 0; // cap of this lane's arena, fixed at first use (pipeline.cpp)
        // (batch size, items per arena chunk) of the last operations that walked a batch in chunks (pipeline.cpp plan_chunk;
        // sealhip_debug_chunk_log): lets a caller that verifies its results pick the items at the chunk boundaries
        std::vector<std::pair<std::size_t, std::size_t>> chunk_log;
        std::recursive_mutex busy; // held for the duration of an operation (a graph may be launched from another thread)
        ~Lane();
    };
    struct LanePool
    {
        std::mutex mu;
        int device = -1;
        unsigned long long id = 0; // serial number, unique per process
        std::vector<std::unique_ptr<Lane>> all;
        std::vector<Lane *> idle; // lanes whose thread has exited
        Lane *take();
        void give(Lane *lane);
    };

    struct Engine
    {
        // parameters
        int scheme = 0, logn = 0, n_key = 0, nsp = 0, k_first = 0;
        std::size_t n = 0;
        u64 t = 0;
        bool mode_strict = false;
        bool use_half_kernel = true; // single-pass NTT kernels for logn >= 14 (the tiled one-pass kernel serves logn <= 13)
        int device = -1; // -1: host-only
        std::vector<u64> key_moduli, aux_primes;
        std::vector<HostNttTables> tables; // per prime id
        // device
        std::shared_ptr<LanePool> lanes;
        Lane &lane() const; // the calling thread's lane of this context (created on first use)
        // Device-side failures are reported through the lane's sticky flag (Lane::h_fault): kernels store to it, every entry
        // point that makes results host-visible reads it after its stream synchronisation (sync_and_check) and fails with
        // E_UNEXPECTED. all_lanes: every lane is synchronised and every lane's flag is read (and cleared).
        void sync_and_check(bool all_lanes = false) const;
        void check_fault() const; // the calling thread's lane alone, after a synchronisation the caller has done itself
        // debug hooks of the NTT hand-off (sealhip_debug_ntt_handoff): spin limit of the sibling wait and
        // suppression of the "finished reading" signal, to drive the failure path in the tests
        unsigned ntt_spin_limit = 1u << 24;
        bool ntt_suppress_signal = false;
        // bumps when a key-switch key is destroyed (graphs embed key pointers); read by other threads' graph launches
        std::atomic<unsigned long long> key_generation{ 0 };
        PrimeDev *d_primes = nullptr;
        std::vector<void *> owned;
        std::map<int, std::unique_ptr<LevelTools>> levels;
        std::map<std::uint32_t, std::uint32_t *> galois_tables; // elt -> device table (galois.cpp:18-47)
        // parms_id of level k (k = n_key: key level), registered by the binding (SEALContext computes them with Blake2,
        // encryptionparams.cpp:132-166); used by the wire-format loader to find a ciphertext's level
        std::map<int, std::array<std::uint64_t, 4>> parms_ids;
        // CKKSEncoder tables (ckks.cpp:37-76): slot map + its inverse (2N words), roots and inverse roots (N complex each)
        std::uint32_t *d_ckks_map = nullptr;
        double *d_ckks_roots = nullptr, *d_ckks_inv_roots = nullptr;
        std::map<int, CkksDecodeDev *> ckks_decode;
        std::map<int, int> total_bits; // significant bits of q_0...q_{k-1} (context.cpp:178)
        void ckks_tables();
        const CkksDecodeDev *ckks_decode_consts(int k);
        int total_coeff_modulus_bit_count(int k);
        int plain_prime = -1;                  // prime id of the plain modulus when batching is possible (context.cpp:262-275)
        std::uint32_t *d_batch_map = nullptr;  // BatchEncoder::matrix_reps_index_map_ (batchencoder.cpp:70-94)
        const std::uint32_t *batch_map();
        mutable std::mutex mu; // lazily built shared tables (levels, Galois tables, encoder tables, parms_ids)
        // profiler (per lane)
        void prof_begin(const char *tag, double units) const;
        void prof_end() const;
        // per-row tickets of the single-pass forward NTT (zeroed for the launch, stream-ordered), lane-owned
        unsigned *ntt_tickets(std::size_t nrows) const;

        ~Engine();
        int n_primes() const
        {
            return static_cast<int>(tables.size());
        }
        LevelTools &level(int k);       // builds host + device constants on first use
        LevelTools &level_host(int k);  // host constants only
        const std::uint32_t *galois_table(std::uint32_t elt);
        void ws_reset() const
        {
            Lane &l = lane();
            l.ws_used = l.ws_floor;
        }
        u64 *ws_alloc(std::size_t words) const;
        void ws_reserve(std::size_t bytes) const;
        RowMap map_for(int k, unsigned base);
        int rows_for(int k, unsigned base);
    };

    struct HipError : std::runtime_error
    {
        hipError_t code;
        HipError(hipError_t c, const char *what) : std::runtime_error(what), code(c)
        {}
    };
#define SEALHIP_CHECK(expr)                                                                          \
    do                                                                                               \
    {                                                                                                \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            throw ::sealhip::HipError(_e, (std::string(#expr) + ": " + hipGetErrorString(_e)).c_str()); \
    } while (0)

    struct ProfScope
    {
        const Engine &e;
        ProfScope(const Engine &eng, const char *tag, double units) : e(eng)
        {
            if (e.lane().prof_on)
                e.prof_begin(tag, units);
        }
        ~ProfScope()
        {
            if (e.lane().prof_on)
                e.prof_end();
        }
    };

    // ---- launchers (each enqueues on e.stream) ----
    hipError_t ntt_init_kernels();
    // measured arithmetic ceiling of a butterfly sequence (ntt.hip butterfly_rate_kernel): butterflies per second
    hipError_t ntt_butterfly_rate(const Engine &e, int kind, int prime_id, double *butterflies_per_s);
    hipError_t launch_ntt(const Engine &e, u64 *data, std::size_t nrows, const RowMap &map, bool inverse, int flags);
    // forward transform whose input rows are gathered from elsewhere (only when ntt_can_gather(e))
    bool ntt_can_gather(const Engine &e);
    bool ntt_strict_top_done_ok(const Engine &e, const RowMap &map); // STRICT: kNttTopDone needs the dense forward schedule
    hipError_t launch_ntt_gather(const Engine &e, u64 *data, std::size_t nrows, const RowMap &map, const NttSource &src,
                                 int flags);

    enum class PolyOp
    {
        Dyadic,
        Add,
        Sub,
        Negate,
        Scalar,
        Mod63 // modulo_poly_coeffs_63, polyarithsmallmod.h:98-120
    };
    hipError_t launch_poly_op(const Engine &e, PolyOp op, const u64 *a, const u64 *b, u64 scalar, u64 *r,
                              std::size_t nrows, const RowMap &map);
    // out[I] = sum_{i1+i2=I} a[i1] (.) b[i2]  over rows of `map` (evaluator.cpp:376-420 / 493-520)
    hipError_t launch_tensor_product(const Engine &e, const u64 *a, int sa, std::size_t a_stride, const u64 *b, int sb,
                                     std::size_t b_stride, u64 *out, std::size_t out_stride, std::size_t count,
                                     const RowMap &map, bool square = false);
    hipError_t launch_fill_rows(const Engine &e, u64 *dst, const u64 *row_values, int rows, std::size_t count);
    hipError_t launch_copy_rows(const Engine &e, const u64 *src, std::size_t src_poly_stride, u64 *dst,
                                std::size_t dst_poly_stride, std::size_t npolys, int rows);

    // RNSTool kernels; item strides are in words
    hipError_t launch_fastbconv_m_tilde(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in,
                                        std::size_t in_stride, u64 *out, std::size_t out_stride, std::size_t count);
    hipError_t launch_sm_mrq(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in, std::size_t in_stride,
                             u64 *out, std::size_t out_stride, std::size_t count);
    // fused fastbconv_m_tilde + sm_mrq: in k rows -> out |Bsk| rows
    // top_layer: also apply the forward NTT's top layer to the rows written (exact-k instances only; the caller then
    // transforms them with kNttTopDone). -> bfv_lift_can_apply_top tells whether this level has such an instance.
    bool bfv_lift_can_apply_top(const Engine &e, const RnsDev &h);
    hipError_t launch_bfv_lift(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in, std::size_t in_stride,
                               u64 *out, std::size_t out_stride, std::size_t count, bool top_layer = false);
    hipError_t launch_fast_floor(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in,
                                 std::size_t in_stride, u64 *out, std::size_t out_stride, std::size_t count,
                                 int mul_t);
    hipError_t launch_fastbconv_sk(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in,
                                   std::size_t in_stride, u64 *out, std::size_t out_stride, std::size_t count);
    // fused (x t) + fast_floor + fastbconv_sk: in (k+|Bsk|) rows -> out k rows
    // deferred_top != 0: `in` holds the output of the single-pass inverse kernel WITHOUT its top layer; the kernel
    // applies that layer and the canonicalising subtraction while loading (needs ntt_can_defer_top(e))
    // deferred_top == 2: as 1, and every input word carries the Montgomery factor 2^-64 (launch_intt_tensor)
    hipError_t launch_bfv_floor_sk(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in,
                                   std::size_t in_stride, u64 *out, std::size_t out_stride, std::size_t count,
                                   int deferred_top = 0);
    bool ntt_can_defer_top(const Engine &e, int k);
    // reduce mode 7 available for a gathered launch over the first k ciphertext primes with this special prime?
    bool ntt_can_fuse_moddown(const Engine &e, int k, u64 p_special);
    // inverse NTT whose input rows come from another buffer (single-pass kernels only: ntt_can_gather(e))
    hipError_t launch_intt_from(const Engine &e, u64 *data, const u64 *src, std::size_t src_poly_stride, std::size_t nrows,
                                const RowMap &map, int flags);
    // inverse NTT that forms the (2,2) ciphertext tensor product on load (needs ntt_can_gather(e)); see ntt.hip
    hipError_t launch_intt_tensor(const Engine &e, u64 *data, const u64 *x, std::size_t item_stride, std::size_t poly_stride,
                                  int kb, std::size_t nrows, const RowMap &map, int flags, bool square = false);
    hipError_t launch_divround_bfv(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in,
                                   std::size_t in_stride, u64 *out, std::size_t out_stride, std::size_t count,
                                   int out_rows);
    // `last` points at the (inverse-transformed) last row of item 0; items are last_stride words apart
    hipError_t launch_rescale_pre(const Engine &e, const RnsDev *d, const RnsDev &h, u64 *last,
                                  std::size_t last_stride, u64 *temp, std::size_t temp_stride, std::size_t count);
    hipError_t launch_rescale_post(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in,
                                   std::size_t in_stride, const u64 *temp, std::size_t temp_stride, u64 *out,
                                   std::size_t out_stride, std::size_t count);

    // Galois
    enum class CtLinearOp
    {
        Add,
        Sub,
        Negate,
        MulPlain
    };
    // batches of ciphertexts [count][size][k][N]; b_item_stride only for MulPlain (0 = one plaintext for all)
    hipError_t launch_ct_linear(const Engine &e, CtLinearOp op, const u64 *a, int sa, const u64 *b, int sb,
                                std::size_t b_item_stride, u64 *out, std::size_t count, const RowMap &map);
    hipError_t launch_nonzero_words(const Engine &e, const u64 *data, std::size_t item_stride, std::size_t words,
                                    std::size_t count, unsigned *flags);
    hipError_t launch_nonzero_tail(const Engine &e, const u64 *ct, std::size_t item_words, std::size_t skip_words,
                                   std::size_t count, unsigned *flags);
    hipError_t launch_out_of_range(const Engine &e, const u64 *ct, std::size_t item_words, std::size_t count,
                                   const RowMap &map, unsigned *flags);
    hipError_t launch_plain_lift(const Engine &e, const u64 *plain, std::size_t plain_stride, u64 *out, std::size_t nplains,
                                 const RowMap &map, u64 t);
    hipError_t launch_galois(const Engine &e, const u64 *in, u64 *out, std::size_t nrows, const RowMap &map,
                             std::uint32_t elt, const std::uint32_t *table /* null: coefficient form */);

    // key switch
    // ext is digit-major: ext + j*ext_digit_stride + item*ext_stride + r*N
    hipError_t launch_ks_modup(const Engine &e, const KsDev *d, const KsDev &h, const u64 *coeff,
                               std::size_t coeff_stride, u64 *ext, std::size_t ext_stride,
                               std::size_t ext_digit_stride, std::size_t count, int only_digit);
    hipError_t launch_ks_mac(const Engine &e, const KsDev *d, const KsDev &h, const u64 *target,
                             std::size_t target_stride, const u64 *ext, std::size_t ext_stride,
                             std::size_t ext_digit_stride, const u64 *key, u64 *prod, std::size_t prod_stride,
                             std::size_t count, int j0 = 0, int j1 = -1); // digits [j0, j1) (default: all of the level)
    hipError_t launch_ks_moddown_pre(const Engine &e, const KsDev *d, const KsDev &h, const u64 *prod,
                                     std::size_t prod_stride, u64 *temp, std::size_t temp_stride, std::size_t npolys);
    // BFV only: steps 1-4 of the rescale + the add into the ciphertext in one kernel; top_deferred = the rows of prod
    // come from an inverse NTT launched with kNttDeferTop. c0_src (both mod-down launchers): when set, the ciphertext is
    // not read: component 0 becomes c0_src + result, component 1 the result (apply_galois: the ciphertext the key switch
    // adds into is (galois(c0), 0), evaluator.cpp:1903-1935 -- no copy of c0, no zero fill of c1).
    hipError_t launch_ks_moddown_bfv(const Engine &e, const KsDev *d, const KsDev &h, const u64 *prod,
                                     std::size_t prod_stride, u64 *ct, std::size_t ct_item_stride, std::size_t npolys,
                                     bool top_deferred, const u64 *c0_src = nullptr, std::size_t c0_stride = 0);
    hipError_t launch_ks_moddown_post(const Engine &e, const KsDev *d, const KsDev &h, u64 *prod,
                                      std::size_t prod_stride, const u64 *temp, std::size_t temp_stride, u64 *ct,
                                      std::size_t ct_item_stride, std::size_t npolys, int add_into_ct,
                                      const u64 *c0_src = nullptr, std::size_t c0_stride = 0);

    // ---- batches of separately allocated host ciphertexts (hostbatch.cpp) ----
    struct HostBatchIO
    {
        struct In
        {
            const u64 *const *ptrs; // one host pointer per item
            std::size_t words;      // words per item
        };
        struct Out
        {
            u64 *const *ptrs;
            std::size_t words;
        };
        std::vector<In> in;
        std::vector<Out> out;
        std::size_t tmp_words = 0; // device scratch per item for the chunk function
    };
    // enqueues the operation for m items on the calling thread's lane: d_in[a] / d_out[a] hold the items of array a back to back
    using HostChunkFn = std::function<void(Engine &, const std::vector<u64 *> &d_in, const std::vector<u64 *> &d_out, u64 *d_tmp,
                                           std::size_t m)>;
    void run_host_batch(Engine &e, const HostBatchIO &io, std::size_t count, const HostChunkFn &fn);
    // pins [ptr, ptr + bytes) in place (hipHostRegister) and remembers the range: run_host_batch copies arrays whose items lie
    // in registered ranges straight between the caller's buffers and the device (no staging copy)
    void host_register(const Engine &e, void *ptr, std::size_t bytes);
    void host_unregister(const Engine &e, void *ptr);

    // ---- composed operations (pipeline.cpp) ----
    // c0_src: see launch_ks_moddown_bfv -- the ciphertext is then write-only: (c0_src + result_0, result_1)
    // SURVEY 8(e) latency mode: the digits of ONE key switch split across devices. partial_out: only the inner product over
    // the digits [j0, j1) is formed, reduced to canonical residues and written there (count x 2 x (k + nsp) x N; ct unused).
    // partial_sum: the element-wise sum of every device's partials (words below ranks * p < 2^63; clobbered): reduced, then
    // the key switch continues from there (target, key unused). Modular sums are associative, so every later word is the one
    // the unsplit operation produces.
    struct KsSplit
    {
        int j0 = 0, j1 = 0;
        u64 *partial_out = nullptr;
        u64 *partial_sum = nullptr;
    };
    void op_switch_key(Engine &e, int k, u64 *ct, std::size_t ct_stride, const u64 *target, std::size_t target_stride,
                       std::size_t count, const KSwitchKey &key, const u64 *c0_src = nullptr, std::size_t c0_stride = 0,
                       const KsSplit *split = nullptr);
    void op_modup(Engine &e, int k, int bundle, u64 *ext, std::size_t count);
    void op_bfv_multiply(Engine &e, int k, const u64 *a, int sa, const u64 *b, int sb, std::size_t count, u64 *out);
    void op_ckks_multiply(Engine &e, int k, const u64 *a, int sa, const u64 *b, int sb, std::size_t count, u64 *out);
    // Evaluator::square as its own path (evaluator.cpp:560-770): the operand is lifted / transformed once
    void op_bfv_square(Engine &e, int k, const u64 *a, int sa, std::size_t count, u64 *out);
    void op_ckks_square(Engine &e, int k, const u64 *a, int sa, std::size_t count, u64 *out);
    void op_mod_switch_scale(Engine &e, int k, const u64 *ct, int size, std::size_t count, u64 *out,
                             std::size_t in_item_stride = 0);
    void op_divround_ntt_inplace(Engine &e, int k, u64 *data, std::size_t count);
    void op_rescale_special_inplace(Engine &e, int k, u64 *poly, std::size_t count);
    void op_apply_galois(Engine &e, int k, u64 *ct, std::size_t count, std::uint32_t elt, const KSwitchKey &key);
    void op_multiply_plain(Engine &e, int k, u64 *ct, int size, std::size_t count, const u64 *plain,
                           std::size_t plain_stride);
    // Decryptor::dot_product_ct_sk_array (decryptor.cpp:218-265) and RNSTool::decrypt_scale_and_round
    void op_dot_product_ct_sk(Engine &e, int k, const u64 *ct, int size, std::size_t count, const u64 *sk_powers,
                              bool is_ntt_form, u64 *out);
    hipError_t launch_dot_sk(const Engine &e, const u64 *ct, int size, std::size_t ct_item_stride, const u64 *sk_powers,
                             std::size_t sk_power_stride, u64 *out, std::size_t count, const RowMap &map, int add_c0);
    hipError_t launch_decrypt_scale_and_round(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in, u64 *out,
                                              std::size_t count);

    // ---- encrypt-side arithmetic, scaling variant, BatchEncoder (rlwe.hip) ----
    struct RlweArgs
    {
        u64 *ct;
        std::size_t ct_item_stride, ct_poly_stride; // words
        int polys, rows;                            // polynomials written per item; rows (prime ids 0..rows-1)
        const u64 *x;                               // per-item multiplicand
        std::size_t x_item_stride, x_poly_stride;
        const u64 *y; // multiplicand shared by all items (secret key / public key, NTT form)
        std::size_t y_poly_stride;
        const std::int32_t *e; // small signed samples, N per polynomial
        std::size_t e_item_stride, e_poly_stride;
        int negate;
    };
    struct ScalingArgs
    {
        const u64 *plain;
        std::size_t plain_item_stride;
        u64 *c0;
        std::size_t c0_item_stride;
        int k, sub;
        u64 t, t_cr0, t_cr1, q_mod_t, threshold;
        u64 div[kMaxModuli]; // floor(q / t) mod q_j  (context.cpp:303-321)
    };
    hipError_t launch_rlwe_stage(const Engine &e, int stage, const RlweArgs &a, std::size_t count);
    hipError_t launch_scaling_variant(const Engine &e, const ScalingArgs &a, std::size_t count);
    hipError_t launch_batch_permute(const Engine &e, bool encode, const u64 *in, std::size_t in_item_stride,
                                    std::size_t nvalues, u64 *out, const std::uint32_t *map, std::size_t count,
                                    u64 signed_t = 0);
    // util/rlwe.cpp:204-300: ct[item] = ([-(a*s + e)]_q, a) over key primes 0..rows-1; a_ntt = count x rows x N
    // uniform words in NTT form, e = count x N, sk_ntt = rows x N (row stride N)
    void op_encrypt_zero_symmetric(Engine &e, int rows, bool is_ntt_form, const u64 *a_ntt, const std::int32_t *noise,
                                   const u64 *sk_ntt, std::size_t count, u64 *ct);
    // util/rlwe.cpp:140-202: ct[item][j] = pk[j]*u + e[j]; pk = 2 x rows x N (NTT form), u = count x N, e = count x 2 x N
    void op_encrypt_zero_asymmetric(Engine &e, int rows, bool is_ntt_form, const u64 *pk, const std::int32_t *u,
                                    const std::int32_t *noise, std::size_t count, u64 *ct);
    // util/scalingvariant.cpp:15-92 on the c0 of every item (ct_item_stride words apart)
    void op_scaling_variant(Engine &e, int k, const u64 *plain, std::size_t plain_item_stride, u64 *ct,
                            std::size_t ct_item_stride, std::size_t count, bool sub);
    // batchencoder.cpp:113-154 / :339-376 (needs a prime plain modulus = 1 mod 2N: Engine::plain_prime >= 0)
    void op_batch_encode(Engine &e, const u64 *values, std::size_t nvalues, std::size_t count, u64 *plain, bool is_signed = false);
    void op_batch_decode(Engine &e, const u64 *plain, std::size_t count, u64 *values, bool is_signed = false);

    // CKKSEncoder (ckks_encoder.hip)
    hipError_t launch_ckks_encode_front(const Engine &e, const double *values, std::size_t n_values, std::size_t count,
                                        double n_inv_scale, double *cv, u64 *out, int rows, const std::uint32_t *map,
                                        const double *inv_roots, int *max_bits);
    hipError_t launch_ckks_decode_back(const Engine &e, const u64 *coeff, const CkksDecodeDev *d, int k, std::size_t count,
                                       double inv_scale, double *res, double *values, const std::uint32_t *map,
                                       const double *roots);
    // values: count x n_values complex doubles (device); plain: count x k x N, NTT form
    void op_ckks_encode(Engine &e, int k, const double *values, std::size_t n_values, std::size_t count, double scale,
                        u64 *plain);
    // values out: count x N/2 complex doubles
    void op_ckks_decode(Engine &e, int k, const u64 *plain, std::size_t count, double scale, double *values);

    std::unique_ptr<Engine> make_engine(int scheme, int logn, const u64 *key_moduli, int n_key, int nsp, u64 t,
                                        bool strict, int device);
} // namespace sealhip
