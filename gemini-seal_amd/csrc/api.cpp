// api.cpp -- the C ABI of include/sealhip.h. Conventions follow the reference's C export layer
// (native/src/seal/c/defines.h:34-58, c/evaluator.cpp:39-47): null checks -> E_POINTER, C++ exceptions
// mapped to HRESULTs (invalid_argument -> E_INVALIDARG, logic_error -> COR_E_INVALIDOPERATION,
// everything else -> E_UNEXPECTED), nothing thrown across the boundary.
#include "../../include/sealhip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include <cstring>
#include <new>

#include "blake2xb.hpp"
#include "engine.hpp"

using namespace sealhip;

namespace sealhip
{
    // wire.cpp
    void wire_set_parms_id(Engine &e, int k, const std::uint64_t *id);
    void wire_peek(const void *bytes, std::size_t len, sealhip_ciphertext_info *info);
    void wire_load(Engine &e, const void *bytes, std::size_t len, sealhip_ciphertext_info *info, u64 *dst,
                   std::size_t capacity_words);
    std::size_t wire_save_size(std::uint32_t size, std::uint32_t k, std::size_t n);
    std::size_t wire_save(Engine &e, const sealhip_ciphertext_info &ci, const u64 *src, void *bytes, std::size_t capacity);
    std::uint32_t wire_load_kswitch_key(Engine &e, const void *bytes, std::size_t len, std::uint32_t index, u64 **d_out,
                                        std::size_t *words_out, std::uint64_t *dim1_out);
    std::size_t wire_kswitch_save_size(const Engine &e, const KSwitchKey *const *keys, std::size_t n_slots);
    std::vector<u64> wire_expand_seed(const Engine &e, int rows, const unsigned char *seed_bytes);
    std::size_t wire_save_kswitch_keys(Engine &e, const KSwitchKey *const *keys, std::size_t n_slots, void *bytes,
                                       std::size_t capacity);
} // namespace sealhip

struct sealhip_context
{
    std::unique_ptr<Engine> engine;
};
struct sealhip_kswitch_key
{
    KSwitchKey key;
};
struct sealhip_graph
{
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    Lane *lane = nullptr;                  // the lane it was captured on (its arena addresses are baked in)
    unsigned long long generation = 0;     // Lane::alloc_generation at capture
    unsigned long long key_generation = 0; // Engine::key_generation at capture (key addresses are baked in too)
    unsigned long long pool_id = 0;        // LanePool::id of the context it belongs to (checked at launch)
};

namespace
{
    thread_local std::string g_last_error;

    long fail(long code, const std::string &msg)
    {
        g_last_error = msg;
        return code;
    }

    // Entry points may be called concurrently from several host threads on one context (the reference's Evaluator is
    // re-entrant, evaluator.h:1375-1377): every thread works on its own lane (stream + arena, engine.hpp). An operation
    // holds its lane's lock until it returns; that lock is only ever contended when a graph captured on one thread is
    // launched from another.
    struct OpLock
    {
        Engine *engine;
        std::unique_lock<std::recursive_mutex> lock;
        explicit OpLock(Engine &e) : engine(&e), lock(e.lane().busy)
        {}
    };
    thread_local std::vector<std::unique_ptr<OpLock>> g_locks;
    struct LockScope
    {
        std::size_t depth = g_locks.size();
        ~LockScope()
        {
            while (g_locks.size() > depth)
                g_locks.pop_back();
        }
    };

    // An operation that fails while its lane is capturing a graph leaves the stream in capture mode and the graph
    // half-built: end the capture, drop the graph and say so (the caller must not call _capture_end for it).
    std::string abort_capture_if_any()
    {
        if (g_locks.empty())
            return "";
        Engine &e = *g_locks.back()->engine;
        Lane &l = e.lane();
        if (!l.capturing)
            return "";
        l.capturing = false;
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(l.stream, &g);
        if (g)
            (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return " (the graph capture in progress on this thread was aborted and discarded)";
    }

    template <class F>
    long guarded(F &&body)
    {
        LockScope scope;
        try
        {
            body();
            return SEALHIP_S_OK;
        }
        catch (const HipError &err)
        {
            const std::string note = abort_capture_if_any();
            return fail(err.code == hipErrorOutOfMemory ? SEALHIP_E_OUTOFMEMORY : SEALHIP_E_UNEXPECTED, err.what() + note);
        }
        catch (const std::invalid_argument &err)
        {
            const std::string note = abort_capture_if_any();
            return fail(SEALHIP_E_INVALIDARG, err.what() + note);
        }
        catch (const std::out_of_range &err)
        {
            const std::string note = abort_capture_if_any();
            return fail(SEALHIP_E_INVALIDARG, err.what() + note);
        }
        catch (const std::logic_error &err)
        {
            const std::string note = abort_capture_if_any();
            return fail(SEALHIP_COR_E_INVALIDOPERATION, err.what() + note);
        }
        catch (const std::bad_alloc &)
        {
            const std::string note = abort_capture_if_any();
            return fail(SEALHIP_E_OUTOFMEMORY, "out of host memory" + note);
        }
        catch (const std::exception &err)
        {
            const std::string note = abort_capture_if_any();
            return fail(SEALHIP_E_UNEXPECTED, err.what() + note);
        }
        catch (...)
        {
            const std::string note = abort_capture_if_any();
            return fail(SEALHIP_E_UNEXPECTED, "unknown error" + note);
        }
    }

#define REQUIRE_PTR(p)                                             \
    do                                                             \
    {                                                              \
        if (!(p))                                                  \
            return fail(SEALHIP_E_POINTER, #p " is null");         \
    } while (0)

    Engine &device_engine(sealhip_context *ctx)
    {
        Engine &e = *ctx->engine;
        if (e.device < 0)
            throw std::logic_error("host-only context: there is no CPU fallback, create the context on a HIP device");
        (void)hipGetLastError(); // an error an earlier call already reported must not be attributed to this one's launches
        SEALHIP_CHECK(hipSetDevice(e.device));
        g_locks.push_back(std::make_unique<OpLock>(e));
        return e;
    }

    void check_level(const Engine &e, uint32_t k)
    {
        if (k < 1 || static_cast<int>(k) > e.n_key)
            throw std::invalid_argument("level k out of range");
    }

    void check_launch(hipError_t err, const char *what)
    {
        if (err != hipSuccess)
            throw HipError(err, (std::string(what) + ": " + hipGetErrorString(err)).c_str());
    }

    void ntt_entry(sealhip_context *ctx, uint64_t *data, size_t count, uint32_t k, uint32_t base, bool inverse,
                   int flags)
    {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (base == SEALHIP_BASE_BSK || base == SEALHIP_BASE_KEY)
            (void)e.level(static_cast<int>(k));
        const RowMap map = e.map_for(static_cast<int>(k), base);
        check_launch(launch_ntt(e, reinterpret_cast<u64 *>(data), count * map.rows, map, inverse, flags), "ntt");
    }

    void poly_entry(sealhip_context *ctx, PolyOp op, const uint64_t *a, const uint64_t *b, uint64_t scalar,
                    uint64_t *r, size_t count, uint32_t k, uint32_t base)
    {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        const RowMap map = e.map_for(static_cast<int>(k), base);
        check_launch(launch_poly_op(e, op, reinterpret_cast<const u64 *>(a), reinterpret_cast<const u64 *>(b), scalar,
                                    reinterpret_cast<u64 *>(r), count * map.rows, map),
                     "poly op");
    }

    // GaloisTool::get_elt_from_step (galois.cpp:49-91), generator 5 (util/galois.h:169)
    std::uint32_t host_galois_elt_from_step(std::size_t n_, int step)
    {
        const uint32_t n = static_cast<uint32_t>(n_);
        const uint64_t m = static_cast<uint64_t>(n) * 2;
        if (step == 0)
            return static_cast<uint32_t>(m - 1);
        const bool negative = step < 0;
        const uint32_t pos = static_cast<uint32_t>(negative ? -static_cast<int64_t>(step) : step);
        if (pos >= (n >> 1))
            throw std::invalid_argument("step count too large");
        uint32_t s = negative ? (n >> 1) - pos : pos;
        uint64_t elt = 1;
        while (s--)
            elt = (elt * 5) & (m - 1);
        return static_cast<uint32_t>(elt);
    }

    LevelTools &bfv_level(Engine &e, uint32_t k)
    {
        check_level(e, k);
        if (e.scheme != 1)
            throw std::logic_error("unsupported operation for scheme type");
        return e.level(static_cast<int>(k));
    }
} // namespace

extern "C" {

const char *sealhip_last_error_string(void)
{
    return g_last_error.c_str();
}

long sealhip_num_devices(int32_t *count)
{
    REQUIRE_PTR(count);
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    *count = err == hipSuccess ? n : 0;
    return SEALHIP_S_OK;
}

long sealhip_context_create(const sealhip_params *params, sealhip_context **out)
{
    REQUIRE_PTR(params);
    REQUIRE_PTR(out);
    REQUIRE_PTR(params->key_moduli);
    *out = nullptr;
    return guarded([&] {
        if (params->mode != SEALHIP_MODE_PARITY && params->mode != SEALHIP_MODE_STRICT)
            throw std::invalid_argument("unknown mode");
        if (params->device >= 0)
        {
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess || params->device >= n)
                throw HipError(hipErrorNoDevice, "no such HIP device (the engine has no CPU fallback)");
        }
        auto ctx = std::make_unique<sealhip_context>();
        ctx->engine = make_engine(static_cast<int>(params->scheme), static_cast<int>(params->log_n),
                                  reinterpret_cast<const u64 *>(params->key_moduli),
                                  static_cast<int>(params->n_key_moduli), static_cast<int>(params->n_special_primes),
                                  params->plain_modulus, params->mode == SEALHIP_MODE_STRICT, params->device);
        *out = ctx.release();
    });
}

long sealhip_context_destroy(sealhip_context *ctx)
{
    REQUIRE_PTR(ctx);
    return guarded([&] { delete ctx; });
}

long sealhip_context_first_level(const sealhip_context *ctx, uint32_t *k_first)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(k_first);
    *k_first = static_cast<uint32_t>(ctx->engine->k_first);
    return SEALHIP_S_OK;
}

long sealhip_context_bsk_size(sealhip_context *ctx, uint32_t k, uint32_t *bsk_size)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(bsk_size);
    return guarded([&] {
        Engine &e = *ctx->engine;
        check_level(e, k);
        if (e.scheme != 1)
            throw std::logic_error("base Bsk exists only for BFV");
        *bsk_size = static_cast<uint32_t>(e.level_host(static_cast<int>(k)).map_bsk.rows);
    });
}

namespace
{
    void adopt_stream(Engine &e, hipStream_t stream, bool own)
    {
        Lane &l = e.lane();
        if (l.capturing)
            throw std::logic_error("a graph capture is in progress on this thread");
        SEALHIP_CHECK(hipStreamSynchronize(l.stream));
        if (l.own_stream)
            SEALHIP_CHECK(hipStreamDestroy(l.stream));
        l.stream = stream;
        l.own_stream = own;
    }
} // namespace

long sealhip_set_stream(sealhip_context *ctx, void *hip_stream)
{
    REQUIRE_PTR(ctx);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        if (hip_stream)
            adopt_stream(e, static_cast<hipStream_t>(hip_stream), false);
        else if (!e.lane().own_stream)
        {
            hipStream_t s = nullptr;
            SEALHIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            adopt_stream(e, s, true);
        }
    });
}

long sealhip_use_default_stream(sealhip_context *ctx)
{
    REQUIRE_PTR(ctx);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        adopt_stream(e, nullptr, false); // the legacy NULL stream: ordered against every blocking stream of the device
    });
}

long sealhip_synchronize(sealhip_context *ctx)
{
    REQUIRE_PTR(ctx);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        e.sync_and_check(true);
    });
}

long sealhip_context_lane_count(sealhip_context *ctx, uint32_t *lanes)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(lanes);
    return guarded([&] {
        Engine &e = *ctx->engine;
        *lanes = 0;
        if (e.lanes)
        {
            std::lock_guard<std::mutex> lock(e.lanes->mu);
            *lanes = static_cast<uint32_t>(e.lanes->all.size());
        }
    });
}

long sealhip_debug_ntt_handoff(sealhip_context *ctx, uint32_t spin_limit, int32_t suppress_signal)
{
    REQUIRE_PTR(ctx);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        e.sync_and_check(true);
        e.ntt_spin_limit = spin_limit ? spin_limit : (1u << 24);
        e.ntt_suppress_signal = suppress_signal != 0;
    });
}

long sealhip_debug_chunk_log(sealhip_context *ctx, size_t *count_chunk_pairs, size_t capacity_pairs, size_t *written)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(count_chunk_pairs);
    REQUIRE_PTR(written);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        auto &log = e.lane().chunk_log;
        std::size_t n = 0;
        for (; n < log.size() && n < capacity_pairs; n++)
        {
            count_chunk_pairs[2 * n] = log[n].first;
            count_chunk_pairs[2 * n + 1] = log[n].second;
        }
        *written = n;
        log.clear();
    });
}

long sealhip_debug_butterfly_rate(sealhip_context *ctx, uint32_t kind, uint32_t prime_index, double *butterflies_per_s)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(butterflies_per_s);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        e.sync_and_check(true);
        const hipError_t err = ntt_butterfly_rate(e, static_cast<int>(kind), static_cast<int>(prime_index), butterflies_per_s);
        if (err == hipErrorInvalidValue)
            throw std::invalid_argument("butterfly_rate: kind 0..5, a prime of the context (kind 3: one below 2^50)");
        SEALHIP_CHECK(err);
    });
}

long sealhip_malloc(sealhip_context *ctx, size_t bytes, void **dptr)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(dptr);
    return guarded([&] {
        device_engine(ctx);
        SEALHIP_CHECK(hipMalloc(dptr, bytes ? bytes : 1));
    });
}

long sealhip_free(sealhip_context *ctx, void *dptr)
{
    REQUIRE_PTR(ctx);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        SEALHIP_CHECK(hipStreamSynchronize(e.lane().stream));
        SEALHIP_CHECK(hipFree(dptr));
    });
}

long sealhip_memcpy_h2d(sealhip_context *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(dst_dev);
    REQUIRE_PTR(src_host);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        SEALHIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, e.lane().stream));
        SEALHIP_CHECK(hipStreamSynchronize(e.lane().stream));
    });
}

long sealhip_memcpy_d2h(sealhip_context *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(dst_host);
    REQUIRE_PTR(src_dev);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        SEALHIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, e.lane().stream));
        e.sync_and_check(); // results become host-visible here: a failed launch must not return S_OK
    });
}

long sealhip_profile_enable(sealhip_context *ctx, int32_t enable)
{
    REQUIRE_PTR(ctx);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        e.lane().prof_on = enable != 0;
    });
}

long sealhip_profile_fetch(sealhip_context *ctx, char *json, size_t capacity)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(json);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        e.sync_and_check();
        struct Agg
        {
            std::size_t launches = 0;
            double ms = 0, units = 0;
        };
        std::map<std::string, Agg> agg;
        for (ProfRecord &r : e.lane().prof)
        {
            float ms = 0;
            if (r.start && r.stop && hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess)
            {
                Agg &a = agg[r.tag];
                a.launches++;
                a.ms += ms;
                a.units += r.units;
            }
            if (r.start)
                (void)hipEventDestroy(r.start);
            if (r.stop)
                (void)hipEventDestroy(r.stop);
        }
        e.lane().prof.clear();
        std::string out = "{";
        bool first = true;
        for (auto &kv : agg)
        {
            char buf[256];
            std::snprintf(buf, sizeof(buf), "%s\"%s\": {\"launches\": %zu, \"ms\": %.6f, \"units\": %.0f}",
                          first ? "" : ", ", kv.first.c_str(), kv.second.launches, kv.second.ms, kv.second.units);
            out += buf;
            first = false;
        }
        out += "}";
        if (out.size() + 1 > capacity)
            throw std::invalid_argument("capacity too small");
        std::memcpy(json, out.c_str(), out.size() + 1);
    });
}

long sealhip_debug_ntt_table(sealhip_context *ctx, uint32_t prime_index, uint32_t kind, uint64_t *out_host,
                             size_t capacity)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(out_host);
    return guarded([&] {
        Engine &e = *ctx->engine;
        if (static_cast<int>(prime_index) >= e.n_primes() || kind > 3)
            throw std::invalid_argument("table index out of range");
        if (capacity < e.n)
            throw std::invalid_argument("capacity too small");
        const HostNttTables &tb = e.tables[prime_index];
        if (tb.fwd.empty())
            throw std::invalid_argument("this prime has no NTT tables");
        const std::vector<u64> v = tb.reference_table(static_cast<int>(kind));
        std::memcpy(out_host, v.data(), v.size() * sizeof(u64));
    });
}

long sealhip_debug_rns_constants(sealhip_context *ctx, uint32_t k, uint32_t which, uint64_t *out_host,
                                 size_t capacity, size_t *written)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(out_host);
    REQUIRE_PTR(written);
    return guarded([&] {
        Engine &e = *ctx->engine;
        check_level(e, k);
        const HostRnsTool &r = *e.level_host(static_cast<int>(k)).host_rns;
        std::vector<u64> v;
        switch (which)
        {
        case 0: v = r.Bsk; break;
        case 1: v = r.inv_prod_q_mod_Bsk; break;
        case 2: v = r.prod_q_mod_Bsk; break;
        case 3: v = r.inv_m_tilde_mod_Bsk; break;
        case 4: v = r.prod_B_mod_q; break;
        case 5: v = r.inv_q_last_mod_q; break;
        case 6: v = { r.inv_prod_q_mod_m_tilde, r.inv_prod_B_mod_m_sk, r.m_sk, r.gamma }; break;
        case 7: v = r.q_to_Bsk.matrix; break;
        case 8: v = r.B_to_q.matrix; break;
        case 9: v = r.q_to_Bsk.inv_punct; break;
        case 10: v = r.B_to_q.inv_punct; break;
        case 11: v = r.q_to_m_tilde.matrix; break;
        case 12: v = r.B_to_m_sk.matrix; break;
        default: throw std::invalid_argument("unknown constant selector");
        }
        if (capacity < v.size())
            throw std::invalid_argument("capacity too small");
        std::memcpy(out_host, v.data(), v.size() * sizeof(u64));
        *written = v.size();
    });
}

/* ---------------------------------------------------------------- NTT */
long sealhip_ntt_negacyclic_harvey_lazy(sealhip_context *ctx, uint64_t *data, size_t count, uint32_t k, uint32_t base)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(data);
    return guarded([&] { ntt_entry(ctx, data, count, k, base, false, 0); });
}
long sealhip_ntt_negacyclic_harvey(sealhip_context *ctx, uint64_t *data, size_t count, uint32_t k, uint32_t base)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(data);
    return guarded([&] { ntt_entry(ctx, data, count, k, base, false, kNttCanonical); });
}
long sealhip_inverse_ntt_negacyclic_harvey_lazy(sealhip_context *ctx, uint64_t *data, size_t count, uint32_t k,
                                                uint32_t base)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(data);
    return guarded([&] { ntt_entry(ctx, data, count, k, base, true, 0); });
}
long sealhip_inverse_ntt_negacyclic_harvey(sealhip_context *ctx, uint64_t *data, size_t count, uint32_t k,
                                           uint32_t base)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(data);
    return guarded([&] { ntt_entry(ctx, data, count, k, base, true, kNttCanonical); });
}

/* ---------------------------------------------------------------- coefficient-wise */
long sealhip_dyadic_product_coeffmod(sealhip_context *ctx, const uint64_t *a, const uint64_t *b, size_t count,
                                     uint32_t k, uint32_t base, uint64_t *result)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(a);
    REQUIRE_PTR(b);
    REQUIRE_PTR(result);
    return guarded([&] { poly_entry(ctx, PolyOp::Dyadic, a, b, 0, result, count, k, base); });
}
long sealhip_multiply_poly_scalar_coeffmod(sealhip_context *ctx, const uint64_t *a, size_t count, uint32_t k,
                                           uint32_t base, uint64_t scalar, uint64_t *result)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(a);
    REQUIRE_PTR(result);
    return guarded([&] { poly_entry(ctx, PolyOp::Scalar, a, a, scalar, result, count, k, base); });
}
long sealhip_add_poly_coeffmod(sealhip_context *ctx, const uint64_t *a, const uint64_t *b, size_t count, uint32_t k,
                               uint32_t base, uint64_t *result)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(a);
    REQUIRE_PTR(b);
    REQUIRE_PTR(result);
    return guarded([&] { poly_entry(ctx, PolyOp::Add, a, b, 0, result, count, k, base); });
}
long sealhip_sub_poly_coeffmod(sealhip_context *ctx, const uint64_t *a, const uint64_t *b, size_t count, uint32_t k,
                               uint32_t base, uint64_t *result)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(a);
    REQUIRE_PTR(b);
    REQUIRE_PTR(result);
    return guarded([&] { poly_entry(ctx, PolyOp::Sub, a, b, 0, result, count, k, base); });
}
long sealhip_negate_poly_coeffmod(sealhip_context *ctx, const uint64_t *a, size_t count, uint32_t k, uint32_t base,
                                  uint64_t *result)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(a);
    REQUIRE_PTR(result);
    return guarded([&] { poly_entry(ctx, PolyOp::Negate, a, a, 0, result, count, k, base); });
}

/* ---------------------------------------------------------------- RNSTool */
long sealhip_fastbconv_m_tilde(sealhip_context *ctx, uint32_t k, const uint64_t *in, size_t count, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(in);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        LevelTools &lt = bfv_level(e, k);
        check_launch(launch_fastbconv_m_tilde(e, lt.d_rns, lt.h_rns, reinterpret_cast<const u64 *>(in), k * e.n,
                                              reinterpret_cast<u64 *>(out), (lt.h_rns.nB + 1) * e.n, count),
                     "fastbconv_m_tilde");
    });
}
long sealhip_sm_mrq(sealhip_context *ctx, uint32_t k, const uint64_t *in, size_t count, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(in);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        LevelTools &lt = bfv_level(e, k);
        check_launch(launch_sm_mrq(e, lt.d_rns, lt.h_rns, reinterpret_cast<const u64 *>(in), (lt.h_rns.nB + 1) * e.n,
                                   reinterpret_cast<u64 *>(out), lt.h_rns.nB * e.n, count),
                     "sm_mrq");
    });
}
long sealhip_fast_floor(sealhip_context *ctx, uint32_t k, const uint64_t *in, size_t count, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(in);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        LevelTools &lt = bfv_level(e, k);
        check_launch(launch_fast_floor(e, lt.d_rns, lt.h_rns, reinterpret_cast<const u64 *>(in),
                                       (k + lt.h_rns.nB) * e.n, reinterpret_cast<u64 *>(out), lt.h_rns.nB * e.n, count,
                                       0),
                     "fast_floor");
    });
}
long sealhip_fastbconv_sk(sealhip_context *ctx, uint32_t k, const uint64_t *in, size_t count, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(in);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        LevelTools &lt = bfv_level(e, k);
        check_launch(launch_fastbconv_sk(e, lt.d_rns, lt.h_rns, reinterpret_cast<const u64 *>(in), lt.h_rns.nB * e.n,
                                         reinterpret_cast<u64 *>(out), k * e.n, count),
                     "fastbconv_sk");
    });
}
long sealhip_divide_and_round_q_last_inplace(sealhip_context *ctx, uint32_t k, uint64_t *data, size_t count)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(data);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (k < 2)
            throw std::invalid_argument("divide_and_round_q_last needs at least two primes");
        LevelTools &lt = e.level(static_cast<int>(k));
        u64 *p = reinterpret_cast<u64 *>(data);
        check_launch(launch_divround_bfv(e, lt.d_rns, lt.h_rns, p, k * e.n, p, k * e.n, count, static_cast<int>(k)),
                     "divide_and_round_q_last");
    });
}
long sealhip_divide_and_round_q_last_ntt_inplace(sealhip_context *ctx, uint32_t k, uint64_t *data, size_t count)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(data);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        op_divround_ntt_inplace(e, static_cast<int>(k), reinterpret_cast<u64 *>(data), count);
    });
}

/* ---------------------------------------------------------------- Galois */
long sealhip_galois_elt_from_step(const sealhip_context *ctx, int32_t step, uint32_t *galois_elt)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(galois_elt);
    return guarded([&] {
        *galois_elt = host_galois_elt_from_step(ctx->engine->n, step);
    });
}

static long galois_entry(sealhip_context *ctx, const uint64_t *in, size_t count, uint32_t k, uint32_t galois_elt,
                         uint64_t *out, bool ntt_form)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(in);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (in == out)
            throw std::invalid_argument("result cannot point to the same value as operand"); // galois.cpp:156-159
        if (!(galois_elt & 1) || galois_elt >= 2 * e.n)
            throw std::invalid_argument("Galois element is not valid");
        const RowMap map = e.map_for(static_cast<int>(k), SEALHIP_BASE_Q);
        const uint32_t *table = ntt_form ? e.galois_table(galois_elt) : nullptr;
        check_launch(launch_galois(e, reinterpret_cast<const u64 *>(in), reinterpret_cast<u64 *>(out), count * k, map,
                                   galois_elt, table),
                     "apply_galois");
    });
}
long sealhip_apply_galois(sealhip_context *ctx, const uint64_t *in, size_t count, uint32_t k, uint32_t galois_elt,
                          uint64_t *out)
{
    return galois_entry(ctx, in, count, k, galois_elt, out, false);
}
long sealhip_apply_galois_ntt(sealhip_context *ctx, const uint64_t *in, size_t count, uint32_t k, uint32_t galois_elt,
                              uint64_t *out)
{
    return galois_entry(ctx, in, count, k, galois_elt, out, true);
}

/* ---------------------------------------------------------------- key switch */
long sealhip_kswitch_key_load(sealhip_context *ctx, const uint64_t *key, uint32_t n_digits, int32_t from_host,
                              sealhip_kswitch_key **out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(key);
    REQUIRE_PTR(out);
    *out = nullptr;
    return guarded([&] {
        Engine &e = device_engine(ctx);
        const uint32_t max_digits = static_cast<uint32_t>((e.k_first + e.nsp - 1) / e.nsp); // keygenerator.cpp:334-336
        if (n_digits == 0 || n_digits > max_digits)
            throw std::invalid_argument("kswitch_keys is not valid for encryption parameters");
        auto k = std::make_unique<sealhip_kswitch_key>();
        k->key.n_digits = n_digits;
        k->key.words = static_cast<std::size_t>(n_digits) * 2 * e.n_key * e.n;
        SEALHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&k->key.d_data), k->key.words * sizeof(u64)));
        hipError_t err = hipMemcpyAsync(k->key.d_data, key, k->key.words * sizeof(u64),
                                        from_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, e.lane().stream);
        if (err == hipSuccess)
            err = hipStreamSynchronize(e.lane().stream);
        if (err != hipSuccess)
        {
            (void)hipFree(k->key.d_data);
            throw HipError(err, hipGetErrorString(err));
        }
        *out = k.release();
    });
}
long sealhip_kswitch_key_destroy(sealhip_context *ctx, sealhip_kswitch_key *key)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(key);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        e.sync_and_check(true); // any thread's lane may still be reading the key
        e.key_generation++;     // graphs captured earlier may embed this key's address: they are stale now
        SEALHIP_CHECK(hipFree(key->key.d_data));
        delete key;
    });
}
long sealhip_modup_rns(sealhip_context *ctx, uint32_t k, uint32_t src_bundle_index, uint64_t *ext, size_t count)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ext);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        op_modup(e, static_cast<int>(k), static_cast<int>(src_bundle_index), reinterpret_cast<u64 *>(ext), count);
    });
}
long sealhip_rescale_special_rns_inplace(sealhip_context *ctx, uint32_t k, uint64_t *poly, size_t count)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(poly);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        op_rescale_special_inplace(e, static_cast<int>(k), reinterpret_cast<u64 *>(poly), count);
    });
}
long sealhip_switch_key_inplace(sealhip_context *ctx, uint32_t k, uint64_t *ct, const uint64_t *target, size_t count,
                                const sealhip_kswitch_key *key)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(target);
    REQUIRE_PTR(key);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        const std::size_t poly = static_cast<std::size_t>(k) * e.n;
        op_switch_key(e, static_cast<int>(k), reinterpret_cast<u64 *>(ct), 2 * poly,
                      reinterpret_cast<const u64 *>(target), poly, count, key->key);
    });
}

/* SURVEY 8(e) "latency mode": the decomposition digits of ONE key switch split across devices (one process per GPU; the
   caller sums the partials with an all-reduce -- RCCL over xGMI -- between the two calls). */
long sealhip_switch_key_partial(sealhip_context *ctx, uint32_t k, const uint64_t *target, size_t count,
                                const sealhip_kswitch_key *key, uint32_t digit_begin, uint32_t digit_end, uint64_t *partial)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(target);
    REQUIRE_PTR(key);
    REQUIRE_PTR(partial);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        const std::size_t poly = static_cast<std::size_t>(k) * e.n;
        KsSplit split;
        split.j0 = static_cast<int>(digit_begin);
        split.j1 = static_cast<int>(digit_end);
        split.partial_out = reinterpret_cast<u64 *>(partial);
        op_switch_key(e, static_cast<int>(k), nullptr, 2 * poly, reinterpret_cast<const u64 *>(target), poly, count, key->key,
                      nullptr, 0, &split);
    });
}

long sealhip_switch_key_finish(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint64_t *partial_sum, size_t count,
                               uint32_t n_partials)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(partial_sum);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        // the sum of n_partials canonical residues is reduced with barrett_reduce_63 (words below 2^63): refuse a world
        // size that could pass it (61-bit key primes: more than 4 partials) instead of wrapping silently
        if (n_partials < 1)
            throw std::invalid_argument("n_partials: at least one partial was summed");
        for (u64 p : e.key_moduli)
            if (static_cast<unsigned __int128>(n_partials) * (p - 1) >= (static_cast<unsigned __int128>(1) << 63))
                throw std::invalid_argument("n_partials * max(key prime) reaches 2^63: reduce the partials before summing more of them");
        const std::size_t poly = static_cast<std::size_t>(k) * e.n;
        KsSplit split;
        split.partial_sum = reinterpret_cast<u64 *>(partial_sum);
        op_switch_key(e, static_cast<int>(k), reinterpret_cast<u64 *>(ct), 2 * poly, nullptr, poly, count, KSwitchKey{}, nullptr, 0,
                      &split);
    });
}

long sealhip_kswitch_digits(sealhip_context *ctx, uint32_t k, uint32_t *digits)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(digits);
    return guarded([&] {
        Engine &e = *ctx->engine;
        check_level(e, k);
        if (static_cast<int>(k) > e.k_first)
            throw std::invalid_argument("key switching needs a ciphertext level");
        *digits = static_cast<uint32_t>((static_cast<int>(k) + e.nsp - 1) / e.nsp); // keygenerator.cpp:334-336
    });
}

namespace
{
    // Transparency as a flag output (sealhip_transparency_sink): an Evaluator entry clears the flags of its batch, then either
    // lets the final kernel of the operation write them (fused: multiply, square, relinearize, apply_galois) or runs the
    // read pass over its result (the remaining entries).
    // Order of effects: the constructor VALIDATES the capacity and nothing else, so an entry can build its
    // scope before its first launch and a call rejected for any reason leaves both the data and the caller's flags as they
    // were. The flags are cleared where they start to be written: by begin() (fused entries call it after their argument
    // checks, right before the operation), by arm() (relinearize: before its last key switch) or by read_pass().
    struct SinkScope
    {
        Engine &e;
        Lane &l;
        bool on;
        size_t count;
        bool cleared = false;
        SinkScope(Engine &eng, size_t n) : e(eng), l(eng.lane()), on(l.tsink != nullptr), count(n)
        {
            if (on && count > l.tsink_cap)
                throw std::invalid_argument("the transparency sink is smaller than this batch");
        }
        void clear()
        {
            if (on && !cleared && count)
                SEALHIP_CHECK(hipMemsetAsync(l.tsink, 0, count * sizeof(unsigned), l.stream));
            cleared = true;
            l.tsink_base = 0;
        }
        void begin() // fused entries (multiply, square, apply_galois): the operation's last kernel writes the flags
        {
            clear();
            if (on)
                l.tsink_cur = l.tsink;
        }
        void arm() // (relinearize: only its last key switch stores the final polynomial 1)
        {
            begin();
        }
        void read_pass(const u64 *result, uint32_t size, size_t poly_words, size_t n)
        {
            clear();
            if (on && size >= 2 && n)
                check_launch(launch_nonzero_tail(e, result, poly_words * size, poly_words, n, l.tsink), "transparency");
        }
        ~SinkScope()
        {
            l.tsink_cur = l.tsink_arm = nullptr;
            l.tsink_base = 0;
        }
    };

    // SEAL_CIPHERTEXT_SIZE_MIN/MAX (util/defines.h:56-57) and the aliasing rule of the raw-buffer form
    void check_multiply_args(Engine &e, uint32_t k, const u64 *a, uint32_t size_a, const u64 *b, uint32_t size_b, const u64 *out)
    {
        check_level(e, k);
        if (size_a < 2 || size_b < 2 || size_a + size_b - 1 > 16)
            throw std::invalid_argument("encrypted1 or encrypted2 is not valid for encryption parameters");
        if (out == a || out == b)
            throw std::invalid_argument("out must not alias an operand");
    }

    // Evaluator::multiply (evaluator.cpp:235-527) on device batches
    void do_multiply(Engine &e, uint32_t k, const u64 *a, uint32_t size_a, const u64 *b, uint32_t size_b, size_t count, u64 *out)
    {
        check_multiply_args(e, k, a, size_a, b, size_b, out);
        if (e.scheme == 1)
            op_bfv_multiply(e, static_cast<int>(k), a, static_cast<int>(size_a), b, static_cast<int>(size_b), count, out);
        else
            op_ckks_multiply(e, static_cast<int>(k), a, static_cast<int>(size_a), b, static_cast<int>(size_b), count, out);
    }

    // relinearize_internal (evaluator.cpp:772-827) on a device batch
    void do_relinearize(Engine &e, uint32_t k, u64 *p, uint32_t size, size_t count, const sealhip_kswitch_key *const *relin_keys,
                        uint32_t n_relin_keys, SinkScope *sink = nullptr)
    {
        check_level(e, k);
        if (size < 2 || size > 16)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (size == 2)
            return; // evaluator.cpp:798-802
        if (!relin_keys || n_relin_keys < size - 2)
            throw std::invalid_argument("not enough relinearization keys"); // :793-796
        const std::size_t poly = static_cast<std::size_t>(k) * e.n;
        // :811-815 -- the target stays the LAST polynomial for every step, exactly like the reference
        for (uint32_t I = 0; I + 2 < size; I++)
        {
            const uint32_t key_power = size - 1 - I;
            const sealhip_kswitch_key *key = relin_keys[key_power - 2];
            if (!key)
                throw std::invalid_argument("not enough relinearization keys");
            if (sink && I + 3 == size)
                sink->arm(); // the last key switch stores the final polynomial 1
            op_switch_key(e, static_cast<int>(k), p, size * poly, p + (size - 1) * poly, size * poly, count, key->key);
        }
    }

    // rotate_internal (evaluator.cpp:1945-2000) on a device batch of size-2 ciphertexts
    void do_rotate(Engine &e, uint32_t k, u64 *ct, size_t count, int32_t steps, const uint32_t *galois_elts,
                   const sealhip_kswitch_key *const *galois_keys, uint32_t n_keys)
    {
        check_level(e, k);
        std::function<void(int)> rotate = [&](int st) {
            if (st == 0)
                return;
            const std::uint32_t elt = host_galois_elt_from_step(e.n, st);
            for (uint32_t i = 0; i < n_keys; i++)
                if (galois_elts[i] == elt)
                {
                    if (!galois_keys[i])
                        throw std::invalid_argument("Galois key not present");
                    op_apply_galois(e, static_cast<int>(k), ct, count, elt, galois_keys[i]->key);
                    return;
                }
            // non-adjacent form, util/numth.h:22-42
            std::vector<int> naf;
            {
                const bool sign = st < 0;
                int value = std::abs(st);
                for (int i = 0; value; i++)
                {
                    const int zi = (value % 2) ? 2 - (value % 4) : 0;
                    value = (value - zi) / 2;
                    if (zi)
                        naf.push_back((sign ? -zi : zi) * (1 << i));
                }
            }
            if (naf.size() == 1)
                throw std::invalid_argument("Galois key not present");
            for (int s : naf)
                if (static_cast<std::size_t>(std::abs(s)) != (e.n >> 1))
                    rotate(s);
        };
        rotate(steps);
    }

    // mod_switch_to_next (evaluator.cpp:996-1036) / rescale_to_next (:1090-1126) on a device batch
    // item_stride (words, 0 = back to back): distance between consecutive ciphertexts of `in` (the *_strided entries)
    void do_level_down(Engine &e, uint32_t k, const u64 *in, uint32_t size, size_t count, u64 *o, bool rescale,
                       size_t item_stride = 0)
    {
        check_level(e, k);
        if (k < 2)
            throw std::invalid_argument("end of modulus switching chain reached");
        if (rescale && e.scheme != 2)
            throw std::invalid_argument("unsupported operation for scheme type"); // evaluator.cpp:1108-1109
        const std::size_t in_poly = static_cast<std::size_t>(k) * e.n, out_poly = static_cast<std::size_t>(k - 1) * e.n;
        if (item_stride != 0 && item_stride < size * in_poly)
            throw std::invalid_argument("item stride smaller than one ciphertext");
        if (rescale || e.scheme == 1)
            op_mod_switch_scale(e, static_cast<int>(k), in, static_cast<int>(size), count, o, item_stride);
        else if (item_stride == 0 || item_stride == size * in_poly)
            // mod_switch_drop_to_next (evaluator.cpp:894-957): keep the first k-1 rows of every polynomial
            check_launch(launch_copy_rows(e, in, in_poly, o, out_poly, count * size, static_cast<int>(k - 1)), "mod_switch_drop");
        else
            for (uint32_t comp = 0; comp < size; comp++)
                check_launch(launch_copy_rows(e, in + comp * in_poly, item_stride, o + comp * out_poly, size * out_poly, count,
                                              static_cast<int>(k - 1)),
                             "mod_switch_drop");
    }
} // namespace

/* ---------------------------------------------------------------- Evaluator level */
long sealhip_evaluator_multiply(sealhip_context *ctx, uint32_t k, const uint64_t *a, uint32_t size_a,
                                const uint64_t *b, uint32_t size_b, size_t count, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(a);
    REQUIRE_PTR(b);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_multiply_args(e, k, reinterpret_cast<const u64 *>(a), size_a, reinterpret_cast<const u64 *>(b), size_b,
                            reinterpret_cast<const u64 *>(out));
        SinkScope sink(e, count);
        sink.begin();
        do_multiply(e, k, reinterpret_cast<const u64 *>(a), size_a, reinterpret_cast<const u64 *>(b), size_b, count,
                    reinterpret_cast<u64 *>(out));
    });
}

long sealhip_evaluator_square(sealhip_context *ctx, uint32_t k, const uint64_t *a, uint32_t size_a, size_t count,
                              uint64_t *out)
{
    // bfv_square / ckks_square (evaluator.cpp:560-770) as their own path: a size-2 operand is lifted / transformed once and
    // c_1 = x_0 x_1 is formed once and added to itself; any other size goes through multiply(a, a) like the reference
    // (:579-583, :720-724). The canonical residues equal multiply(a, a)'s: 2 c0 c1 mod p == c0 c1 + c1 c0 mod p.
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(a);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (size_a < 2 || 2 * size_a - 1 > 16)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (out == a)
            throw std::invalid_argument("out must not alias the operand");
        const u64 *pa = reinterpret_cast<const u64 *>(a);
        SinkScope sink(e, count);
        sink.begin();
        if (e.scheme == 1)
            op_bfv_square(e, static_cast<int>(k), pa, static_cast<int>(size_a), count, reinterpret_cast<u64 *>(out));
        else
            op_ckks_square(e, static_cast<int>(k), pa, static_cast<int>(size_a), count, reinterpret_cast<u64 *>(out));
    });
}

long sealhip_evaluator_relinearize(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size, size_t count,
                                   const sealhip_kswitch_key *const *relin_keys, uint32_t n_relin_keys)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        SinkScope sink(e, count);
        do_relinearize(e, k, reinterpret_cast<u64 *>(ct), size, count, relin_keys, n_relin_keys, &sink);
        if (size == 2) // nothing was done (evaluator.cpp:798-802): the flags describe the input
            sink.read_pass(reinterpret_cast<const u64 *>(ct), 2, static_cast<std::size_t>(k) * e.n, count);
    });
}

/* Evaluator::multiply_many (evaluator.cpp:1180-1255) and exponentiate_inplace (:1257-1288) on device batches of size-2
   ciphertexts: the reference's queue order -- neighbours multiplied left to right, an odd last operand appended, then
   products of products appended until one is left -- every product relinearized. Temporaries come from the stream-ordered
   allocator of the calling thread's lane: nothing here synchronises. */
namespace
{
    void do_multiply_many(Engine &e, uint32_t k, const u64 *const *enc, std::size_t n_enc, size_t count,
                          const sealhip_kswitch_key *const *relin_keys, uint32_t n_relin_keys, u64 *out)
    {
        check_level(e, k);
        if (n_enc == 0)
            throw std::invalid_argument("encrypteds vector must not be empty"); // :1185-1188
        if (e.scheme != 1)
            throw std::logic_error("unsupported scheme"); // :1212-1215
        for (std::size_t i = 0; i < n_enc; i++)
        {
            if (!enc[i])
                throw std::invalid_argument("encrypteds is not valid for encryption parameters");
            if (enc[i] == out)
                throw std::invalid_argument("encrypteds must be different from destination"); // :1193-1199
        }
        const std::size_t poly = static_cast<std::size_t>(k) * e.n, two = count * 2 * poly * sizeof(u64);
        hipStream_t stream = e.lane().stream;
        if (n_enc == 1 || count == 0)
        {
            if (count)
                SEALHIP_CHECK(hipMemcpyAsync(out, enc[0], two, hipMemcpyDeviceToDevice, stream)); // :1218-1222
            return;
        }
        std::vector<void *> owned;
        struct Cleanup
        {
            std::vector<void *> &v;
            hipStream_t s;
            ~Cleanup()
            {
                for (void *p : v)
                    (void)hipFreeAsync(p, s);
            }
        } cleanup{ owned, stream };
        auto product = [&](const u64 *a, const u64 *b) {
            void *wide = nullptr, *narrow = nullptr;
            SEALHIP_CHECK(hipMallocAsync(&wide, count * 3 * poly * sizeof(u64), stream));
            owned.push_back(wide);
            SEALHIP_CHECK(hipMallocAsync(&narrow, two, stream));
            owned.push_back(narrow);
            // (multiply(x, x) and square(x) give the same canonical residues, :1228-1235)
            do_multiply(e, k, a, 2, b, 2, count, static_cast<u64 *>(wide));
            do_relinearize(e, k, static_cast<u64 *>(wide), 3, count, relin_keys, n_relin_keys);
            check_launch(launch_copy_rows(e, static_cast<u64 *>(wide), 3 * poly, static_cast<u64 *>(narrow), 2 * poly, count,
                                          static_cast<int>(2 * k)),
                         "resize");
            return static_cast<const u64 *>(narrow);
        };
        std::vector<const u64 *> queue;
        for (std::size_t i = 0; i + 1 < n_enc; i += 2)
            queue.push_back(product(enc[i], enc[i + 1]));
        if (n_enc & 1)
            queue.push_back(enc[n_enc - 1]);
        for (std::size_t i = 0; i + 1 < queue.size(); i += 2)
            queue.push_back(product(queue[i], queue[i + 1]));
        SEALHIP_CHECK(hipMemcpyAsync(out, queue.back(), two, hipMemcpyDeviceToDevice, stream));
    }
} // namespace

long sealhip_evaluator_multiply_many(sealhip_context *ctx, uint32_t k, const uint64_t *const *encrypteds, uint32_t n_encrypteds,
                                     size_t count, const sealhip_kswitch_key *const *relin_keys, uint32_t n_relin_keys,
                                     uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(out);
    if (n_encrypteds)
        REQUIRE_PTR(encrypteds);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        do_multiply_many(e, k, reinterpret_cast<const u64 *const *>(encrypteds), n_encrypteds, count, relin_keys, n_relin_keys,
                         reinterpret_cast<u64 *>(out));
    });
}

long sealhip_evaluator_exponentiate(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint64_t exponent, size_t count,
                                    const sealhip_kswitch_key *const *relin_keys, uint32_t n_relin_keys, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        if (exponent == 0)
            throw std::invalid_argument("exponent cannot be 0"); // :1275-1278
        if (exponent > (1u << 16))
            throw std::invalid_argument("exponent is too large"); // (the reference would copy the ciphertext `exponent` times)
        if (ct == out && exponent > 1)
            throw std::invalid_argument("out must not alias the operand");
        // :1286-1287 -- multiply_many over `exponent` copies of the ciphertext
        const std::vector<const u64 *> copies(static_cast<std::size_t>(exponent), reinterpret_cast<const u64 *>(ct));
        if (exponent == 1 && ct == out)
            return;
        do_multiply_many(e, k, copies.data(), copies.size(), count, relin_keys, n_relin_keys, reinterpret_cast<u64 *>(out));
    });
}

/* ---------------------------------------------------------------- batches of separately allocated HOST ciphertexts */
long sealhip_evaluator_multiply_host(sealhip_context *ctx, uint32_t k, const uint64_t *const *a, uint32_t size_a,
                                     const uint64_t *const *b, uint32_t size_b, size_t count, uint64_t *const *out,
                                     const sealhip_kswitch_key *const *relin_keys, uint32_t n_relin_keys)
{
    REQUIRE_PTR(ctx);
    if (count)
    {
        REQUIRE_PTR(a);
        REQUIRE_PTR(b);
        REQUIRE_PTR(out);
    }
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (size_a < 2 || size_b < 2 || size_a + size_b - 1 > 16)
            throw std::invalid_argument("encrypted1 or encrypted2 is not valid for encryption parameters");
        const uint32_t dest = size_a + size_b - 1;
        const bool relin = relin_keys != nullptr && dest > 2;
        const std::size_t poly = static_cast<std::size_t>(k) * e.n;
        HostBatchIO io;
        io.in.push_back({ reinterpret_cast<const u64 *const *>(a), size_a * poly });
        io.in.push_back({ reinterpret_cast<const u64 *const *>(b), size_b * poly });
        io.out.push_back({ reinterpret_cast<u64 *const *>(out), (relin ? 2 : dest) * poly });
        io.tmp_words = relin ? dest * poly : 0;
        run_host_batch(e, io, count, [&](Engine &en, const std::vector<u64 *> &d_in, const std::vector<u64 *> &d_out, u64 *tmp,
                                         std::size_t m) {
            if (!relin)
                return do_multiply(en, k, d_in[0], size_a, d_in[1], size_b, m, d_out[0]);
            do_multiply(en, k, d_in[0], size_a, d_in[1], size_b, m, tmp);
            do_relinearize(en, k, tmp, dest, m, relin_keys, n_relin_keys);
            check_launch(launch_copy_rows(en, tmp, dest * poly, d_out[0], 2 * poly, m, static_cast<int>(2 * k)), "resize"); // :819
        });
    });
}

long sealhip_evaluator_relinearize_host(sealhip_context *ctx, uint32_t k, uint64_t *const *ct, uint32_t size, size_t count,
                                        const sealhip_kswitch_key *const *relin_keys, uint32_t n_relin_keys)
{
    REQUIRE_PTR(ctx);
    if (count)
        REQUIRE_PTR(ct);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (size < 2 || size > 16)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (size == 2)
            return;
        const std::size_t poly = static_cast<std::size_t>(k) * e.n;
        HostBatchIO io;
        io.in.push_back({ reinterpret_cast<const u64 *const *>(ct), size * poly });
        io.out.push_back({ reinterpret_cast<u64 *const *>(ct), 2 * poly });
        run_host_batch(e, io, count, [&](Engine &en, const std::vector<u64 *> &d_in, const std::vector<u64 *> &d_out, u64 *,
                                         std::size_t m) {
            do_relinearize(en, k, d_in[0], size, m, relin_keys, n_relin_keys);
            check_launch(launch_copy_rows(en, d_in[0], size * poly, d_out[0], 2 * poly, m, static_cast<int>(2 * k)), "resize");
        });
    });
}

long sealhip_evaluator_rotate_vector_host(sealhip_context *ctx, uint32_t k, uint64_t *const *ct, size_t count, int32_t steps,
                                          const uint32_t *galois_elts, const sealhip_kswitch_key *const *galois_keys,
                                          uint32_t n_keys)
{
    REQUIRE_PTR(ctx);
    if (count)
        REQUIRE_PTR(ct);
    if (n_keys)
    {
        REQUIRE_PTR(galois_elts);
        REQUIRE_PTR(galois_keys);
    }
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        const std::size_t item = 2 * static_cast<std::size_t>(k) * e.n;
        HostBatchIO io;
        io.in.push_back({ reinterpret_cast<const u64 *const *>(ct), item });
        io.out.push_back({ reinterpret_cast<u64 *const *>(ct), item });
        run_host_batch(e, io, count, [&](Engine &en, const std::vector<u64 *> &d_in, const std::vector<u64 *> &d_out, u64 *,
                                         std::size_t m) {
            do_rotate(en, k, d_in[0], m, steps, galois_elts, galois_keys, n_keys);
            SEALHIP_CHECK(hipMemcpyAsync(d_out[0], d_in[0], m * item * sizeof(u64), hipMemcpyDeviceToDevice, en.lane().stream));
        });
    });
}

static long level_down_host(sealhip_context *ctx, uint32_t k, const uint64_t *const *ct, uint32_t size, size_t count,
                            uint64_t *const *out, bool rescale)
{
    REQUIRE_PTR(ctx);
    if (count)
    {
        REQUIRE_PTR(ct);
        REQUIRE_PTR(out);
    }
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (k < 2)
            throw std::invalid_argument("end of modulus switching chain reached");
        if (size < 1 || size > 16)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        HostBatchIO io;
        io.in.push_back({ reinterpret_cast<const u64 *const *>(ct), static_cast<std::size_t>(size) * k * e.n });
        io.out.push_back({ reinterpret_cast<u64 *const *>(out), static_cast<std::size_t>(size) * (k - 1) * e.n });
        run_host_batch(e, io, count, [&](Engine &en, const std::vector<u64 *> &d_in, const std::vector<u64 *> &d_out, u64 *,
                                         std::size_t m) { do_level_down(en, k, d_in[0], size, m, d_out[0], rescale); });
    });
}

long sealhip_evaluator_mod_switch_to_next_host(sealhip_context *ctx, uint32_t k, const uint64_t *const *ct, uint32_t size,
                                               size_t count, uint64_t *const *out)
{
    return level_down_host(ctx, k, ct, size, count, out, false);
}

long sealhip_evaluator_rescale_to_next_host(sealhip_context *ctx, uint32_t k, const uint64_t *const *ct, uint32_t size,
                                            size_t count, uint64_t *const *out)
{
    return level_down_host(ctx, k, ct, size, count, out, true);
}

long sealhip_host_register(sealhip_context *ctx, void *ptr, size_t bytes)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ptr);
    return guarded([&] { host_register(device_engine(ctx), ptr, bytes); });
}

long sealhip_host_unregister(sealhip_context *ctx, void *ptr)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ptr);
    return guarded([&] { host_unregister(device_engine(ctx), ptr); });
}

long sealhip_evaluator_mod_switch_to_next(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size,
                                          size_t count, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        SinkScope sink(e, count);
        do_level_down(e, k, reinterpret_cast<const u64 *>(ct), size, count, reinterpret_cast<u64 *>(out), false);
        sink.read_pass(reinterpret_cast<const u64 *>(out), size, static_cast<std::size_t>(k - 1) * e.n, count);
    });
}

long sealhip_evaluator_mod_switch_to_next_strided(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size,
                                                  size_t ct_item_stride, size_t count, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        SinkScope sink(e, count);
        do_level_down(e, k, reinterpret_cast<const u64 *>(ct), size, count, reinterpret_cast<u64 *>(out), false, ct_item_stride);
        sink.read_pass(reinterpret_cast<const u64 *>(out), size, static_cast<std::size_t>(k - 1) * e.n, count);
    });
}

long sealhip_evaluator_rescale_to_next_strided(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size,
                                               size_t ct_item_stride, size_t count, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        SinkScope sink(e, count);
        do_level_down(e, k, reinterpret_cast<const u64 *>(ct), size, count, reinterpret_cast<u64 *>(out), true, ct_item_stride);
        sink.read_pass(reinterpret_cast<const u64 *>(out), size, static_cast<std::size_t>(k - 1) * e.n, count);
    });
}

long sealhip_evaluator_rescale_to_next(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size,
                                       size_t count, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        SinkScope sink(e, count);
        do_level_down(e, k, reinterpret_cast<const u64 *>(ct), size, count, reinterpret_cast<u64 *>(out), true);
        sink.read_pass(reinterpret_cast<const u64 *>(out), size, static_cast<std::size_t>(k - 1) * e.n, count);
    });
}

long sealhip_evaluator_apply_galois(sealhip_context *ctx, uint32_t k, uint64_t *ct, size_t count,
                                    uint32_t galois_elt, const sealhip_kswitch_key *galois_key)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(galois_key);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        SinkScope sink(e, count);
        sink.begin();
        op_apply_galois(e, static_cast<int>(k), reinterpret_cast<u64 *>(ct), count, galois_elt, galois_key->key);
    });
}

long sealhip_evaluator_transform_to_ntt(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size, size_t count)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    return guarded([&] { ntt_entry(ctx, ct, count * size, k, SEALHIP_BASE_Q, false, kNttCanonical); });
}

long sealhip_evaluator_transform_from_ntt(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size, size_t count)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    return guarded([&] { ntt_entry(ctx, ct, count * size, k, SEALHIP_BASE_Q, true, kNttCanonical); });
}

/* ------------------------------------------------------------------ Evaluator surface beyond the hot path */

long sealhip_evaluator_negate(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size, size_t count,
                              uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (size < 1)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        SinkScope sink(e, count);
        check_launch(launch_ct_linear(e, CtLinearOp::Negate, reinterpret_cast<const u64 *>(ct), static_cast<int>(size),
                                      nullptr, 0, 0, reinterpret_cast<u64 *>(out), count,
                                      e.map_for(static_cast<int>(k), SEALHIP_BASE_Q)),
                     "negate");
        sink.read_pass(reinterpret_cast<const u64 *>(out), size, static_cast<std::size_t>(k) * e.n, count);
    });
}

static long add_sub_entry(sealhip_context *ctx, uint32_t k, const uint64_t *a, uint32_t size_a, const uint64_t *b,
                          uint32_t size_b, size_t count, uint64_t *out, bool sub)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(a);
    REQUIRE_PTR(b);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (size_a < 1 || size_b < 1)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        // in place (out == a) is only well defined when the result is not larger than a (the reference resizes
        // encrypted1 first, evaluator.cpp:131-132; a raw buffer cannot grow)
        if (out == a && size_b > size_a)
            throw std::invalid_argument("in-place result needs a destination of max(size_a, size_b) polynomials");
        SinkScope sink(e, count);
        check_launch(launch_ct_linear(e, sub ? CtLinearOp::Sub : CtLinearOp::Add, reinterpret_cast<const u64 *>(a),
                                      static_cast<int>(size_a), reinterpret_cast<const u64 *>(b), static_cast<int>(size_b), 0,
                                      reinterpret_cast<u64 *>(out), count, e.map_for(static_cast<int>(k), SEALHIP_BASE_Q)),
                     sub ? "sub" : "add");
        sink.read_pass(reinterpret_cast<const u64 *>(out), std::max(size_a, size_b), static_cast<std::size_t>(k) * e.n, count);
    });
}

long sealhip_evaluator_add(sealhip_context *ctx, uint32_t k, const uint64_t *a, uint32_t size_a, const uint64_t *b,
                           uint32_t size_b, size_t count, uint64_t *out)
{
    return add_sub_entry(ctx, k, a, size_a, b, size_b, count, out, false);
}

long sealhip_evaluator_sub(sealhip_context *ctx, uint32_t k, const uint64_t *a, uint32_t size_a, const uint64_t *b,
                           uint32_t size_b, size_t count, uint64_t *out)
{
    return add_sub_entry(ctx, k, a, size_a, b, size_b, count, out, true);
}

long sealhip_evaluator_multiply_plain_ntt(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size, size_t count,
                                          const uint64_t *plain_ntt, size_t plain_stride)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(plain_ntt);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (size < 1)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (plain_stride != 0 && plain_stride < static_cast<size_t>(k) * e.n)
            throw std::invalid_argument("plain_stride is smaller than one plaintext");
        SinkScope sink(e, count); // (capacity checked before ct is overwritten)
        check_launch(launch_ct_linear(e, CtLinearOp::MulPlain, reinterpret_cast<const u64 *>(ct), static_cast<int>(size),
                                      reinterpret_cast<const u64 *>(plain_ntt), 0, plain_stride, reinterpret_cast<u64 *>(ct),
                                      count, e.map_for(static_cast<int>(k), SEALHIP_BASE_Q)),
                     "multiply_plain_ntt");
        sink.read_pass(reinterpret_cast<const u64 *>(ct), size, static_cast<std::size_t>(k) * e.n, count);
    });
}

long sealhip_evaluator_multiply_plain(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size, size_t count,
                                      const uint64_t *plain, size_t plain_stride)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(plain);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (size < 1)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (plain_stride != 0 && plain_stride < e.n)
            throw std::invalid_argument("plain_stride is smaller than one plaintext");
        SinkScope sink(e, count); // (capacity checked before ct is overwritten)
        op_multiply_plain(e, static_cast<int>(k), reinterpret_cast<u64 *>(ct), static_cast<int>(size), count,
                          reinterpret_cast<const u64 *>(plain), plain_stride);
        sink.read_pass(reinterpret_cast<const u64 *>(ct), size, static_cast<std::size_t>(k) * e.n, count);
    });
}

long sealhip_transparency_sink(sealhip_context *ctx, uint32_t *nonzero_flags, size_t capacity)
{
    REQUIRE_PTR(ctx);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        Lane &l = e.lane();
        if (nonzero_flags && capacity == 0)
            throw std::invalid_argument("a transparency sink needs room for at least one ciphertext");
        static_assert(sizeof(unsigned) == sizeof(uint32_t), "flag words");
        l.tsink = reinterpret_cast<unsigned *>(nonzero_flags);
        l.tsink_cap = nonzero_flags ? capacity : 0;
        l.tsink_cur = l.tsink_arm = nullptr;
    });
}

long sealhip_is_transparent(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size, size_t count,
                            uint8_t *transparent)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(transparent);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (count == 0)
            return;
        if (size < 2) // ciphertext.h:474: fewer than SEAL_CIPHERTEXT_SIZE_MIN polynomials
        {
            std::fill(transparent, transparent + count, static_cast<uint8_t>(1));
            return;
        }
        const std::size_t poly_words = static_cast<std::size_t>(k) * e.n;
        e.ws_reserve(e.lane().ws_floor + count * sizeof(unsigned) + 512);
        e.ws_reset();
        unsigned *flags = reinterpret_cast<unsigned *>(e.ws_alloc((count * sizeof(unsigned) + 7) / 8));
        SEALHIP_CHECK(hipMemsetAsync(flags, 0, count * sizeof(unsigned), e.lane().stream));
        check_launch(launch_nonzero_tail(e, reinterpret_cast<const u64 *>(ct), poly_words * size, poly_words, count, flags),
                     "is_transparent");
        std::vector<unsigned> host(count);
        SEALHIP_CHECK(hipMemcpyAsync(host.data(), flags, count * sizeof(unsigned), hipMemcpyDeviceToHost, e.lane().stream));
        e.sync_and_check();
        for (size_t i = 0; i < count; i++)
            transparent[i] = host[i] ? 0 : 1;
    });
}

long sealhip_modulo_poly_coeffs_63(sealhip_context *ctx, const uint64_t *a, size_t count, uint32_t k, uint32_t base,
                                   uint64_t *result)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(a);
    REQUIRE_PTR(result);
    return guarded([&] { poly_entry(ctx, PolyOp::Mod63, a, nullptr, 0, result, count, k, base); });
}

long sealhip_evaluator_rotate_vector(sealhip_context *ctx, uint32_t k, uint64_t *ct, size_t count, int32_t steps,
                                     const uint32_t *galois_elts, const sealhip_kswitch_key *const *galois_keys,
                                     uint32_t n_keys)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    if (n_keys)
    {
        REQUIRE_PTR(galois_elts);
        REQUIRE_PTR(galois_keys);
    }
    return guarded([&] {
        Engine &e = device_engine(ctx);
        // (the NAF fallback applies several automorphisms: the flags are read off the final ciphertext)
        SinkScope sink(e, count);
        do_rotate(e, k, reinterpret_cast<u64 *>(ct), count, steps, galois_elts, galois_keys, n_keys);
        sink.read_pass(reinterpret_cast<const u64 *>(ct), 2, static_cast<std::size_t>(k) * e.n, count);
    });
}

/* ------------------------------------------------------------------ decrypt-side arithmetic */

long sealhip_decryptor_dot_product_ct_sk(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size, size_t count,
                                         const uint64_t *sk_powers_ntt, int32_t is_ntt_form, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(out);
    if (size > 1)
        REQUIRE_PTR(sk_powers_ntt);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (size < 1)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        op_dot_product_ct_sk(e, static_cast<int>(k), reinterpret_cast<const u64 *>(ct), static_cast<int>(size), count,
                             reinterpret_cast<const u64 *>(sk_powers_ntt), is_ntt_form != 0, reinterpret_cast<u64 *>(out));
    });
}

long sealhip_decrypt_scale_and_round(sealhip_context *ctx, uint32_t k, const uint64_t *in, size_t count, uint64_t *out)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(in);
    REQUIRE_PTR(out);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        LevelTools &lt = bfv_level(e, k);
        check_launch(launch_decrypt_scale_and_round(e, lt.d_rns, lt.h_rns, reinterpret_cast<const u64 *>(in),
                                                    reinterpret_cast<u64 *>(out), count),
                     "decrypt_scale_and_round");
    });
}

/* ------------------------------------------------------------------ HIP graphs */

long sealhip_graph_capture_begin(sealhip_context *ctx)
{
    REQUIRE_PTR(ctx);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        if (e.lane().capturing)
            throw std::logic_error("a capture is already in progress on this context");
        if (e.lane().prof_on)
            throw std::logic_error("disable the launch profiler before capturing");
        SEALHIP_CHECK(hipStreamBeginCapture(e.lane().stream, hipStreamCaptureModeRelaxed));
        e.lane().capturing = true;
    });
}

long sealhip_graph_capture_end(sealhip_context *ctx, sealhip_graph **graph)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(graph);
    *graph = nullptr;
    return guarded([&] {
        Engine &e = device_engine(ctx);
        if (!e.lane().capturing)
            throw std::logic_error("no capture in progress");
        e.lane().capturing = false;
        auto g = std::make_unique<sealhip_graph>();
        SEALHIP_CHECK(hipStreamEndCapture(e.lane().stream, &g->graph));
        if (!g->graph)
            throw std::logic_error("the capture was invalidated (an operation allocated or synchronised): run the sequence "
                                   "once before capturing");
        const hipError_t err = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
        if (err != hipSuccess)
        {
            (void)hipGraphDestroy(g->graph);
            throw HipError(err, hipGetErrorString(err));
        }
        g->lane = &e.lane();
        g->generation = e.lane().alloc_generation;
        g->key_generation = e.key_generation.load();
        g->pool_id = e.lanes->id;
        *graph = g.release();
    });
}

long sealhip_graph_launch(sealhip_context *ctx, sealhip_graph *graph)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(graph);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        if (graph->pool_id != e.lanes->id)
            throw std::invalid_argument("the graph was captured on another context");
        // replay on the lane the graph was captured on (its arena is part of the graph), whichever thread asks. The caller's
        // own lane is released first: two threads that launch graphs captured on each other's lanes would otherwise take
        // the two lane locks in opposite order
        Lane &l = *graph->lane;
        if (&l != &e.lane() && !g_locks.empty())
            g_locks.pop_back();
        std::lock_guard<std::recursive_mutex> busy(l.busy);
        if (graph->generation != l.alloc_generation)
            throw std::logic_error("the graph is stale: the workspace was re-allocated by a larger operation after the capture");
        if (graph->key_generation != e.key_generation.load())
            throw std::logic_error("the graph is stale: a key-switch key was destroyed after the capture");
        SEALHIP_CHECK(hipGraphLaunch(graph->exec, l.stream));
    });
}

long sealhip_graph_destroy(sealhip_context *ctx, sealhip_graph *graph)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(graph);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        e.sync_and_check(true);
        (void)hipGraphExecDestroy(graph->exec);
        (void)hipGraphDestroy(graph->graph);
        delete graph;
    });
}

/* ------------------------------------------------------------------ encrypt-side arithmetic */

long sealhip_encrypt_zero_symmetric(sealhip_context *ctx, uint32_t rows, int32_t is_ntt_form, const uint64_t *a_ntt,
                                    const int32_t *noise, const uint64_t *sk_ntt, size_t count, uint64_t *ct)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(a_ntt);
    REQUIRE_PTR(noise);
    REQUIRE_PTR(sk_ntt);
    REQUIRE_PTR(ct);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, rows);
        op_encrypt_zero_symmetric(e, static_cast<int>(rows), is_ntt_form != 0, reinterpret_cast<const u64 *>(a_ntt), noise,
                                  reinterpret_cast<const u64 *>(sk_ntt), count, reinterpret_cast<u64 *>(ct));
    });
}

long sealhip_encrypt_zero_asymmetric(sealhip_context *ctx, uint32_t rows, int32_t is_ntt_form, const uint64_t *pk_ntt,
                                     const int32_t *u, const int32_t *noise, size_t count, uint64_t *ct)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(pk_ntt);
    REQUIRE_PTR(u);
    REQUIRE_PTR(noise);
    REQUIRE_PTR(ct);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, rows);
        op_encrypt_zero_asymmetric(e, static_cast<int>(rows), is_ntt_form != 0, reinterpret_cast<const u64 *>(pk_ntt), u,
                                   noise, count, reinterpret_cast<u64 *>(ct));
    });
}

long sealhip_multiply_add_plain_with_scaling_variant(sealhip_context *ctx, uint32_t k, const uint64_t *plain,
                                                     size_t plain_item_stride, uint64_t *ct, uint32_t size, size_t count,
                                                     int32_t subtract)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(plain);
    REQUIRE_PTR(ct);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (e.scheme != 1)
            throw std::logic_error("unsupported operation for scheme type");
        if (size < 1)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        op_scaling_variant(e, static_cast<int>(k), reinterpret_cast<const u64 *>(plain), plain_item_stride,
                           reinterpret_cast<u64 *>(ct), static_cast<std::size_t>(size) * k * e.n, count, subtract != 0);
    });
}

long sealhip_evaluator_add_plain(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size, size_t count,
                                 const uint64_t *plain, size_t plain_item_stride, int32_t subtract)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(plain);
    REQUIRE_PTR(ct);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (size < 1)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        const std::size_t item = static_cast<std::size_t>(size) * k * e.n;
        if (e.scheme == 1) // evaluator.cpp:1338-1342 / :1412-1416
        {
            op_scaling_variant(e, static_cast<int>(k), reinterpret_cast<const u64 *>(plain), plain_item_stride,
                               reinterpret_cast<u64 *>(ct), item, count, subtract != 0);
            return;
        }
        // CKKS: c_0 +-= plain, both in NTT form (evaluator.cpp:1344-1350 / :1418-1424)
        const RowMap map = e.map_for(static_cast<int>(k), SEALHIP_BASE_Q);
        u64 *c = reinterpret_cast<u64 *>(ct);
        const u64 *p = reinterpret_cast<const u64 *>(plain);
        if (plain_item_stride == static_cast<std::size_t>(k) * e.n)
        {
            // a batch of plaintexts laid out back to back is a batch of size-1 operands of add/sub (evaluator.cpp:131-143)
            check_launch(launch_ct_linear(e, subtract ? CtLinearOp::Sub : CtLinearOp::Add, c, static_cast<int>(size), p, 1, 0, c,
                                          count, map),
                         "add_plain");
            return;
        }
        for (std::size_t i = 0; i < count; i++)
            check_launch(launch_poly_op(e, subtract ? PolyOp::Sub : PolyOp::Add, c + i * item, p + i * plain_item_stride, 0,
                                        c + i * item, k, map),
                         "add_plain");
    });
}

/* ------------------------------------------------------------------ BatchEncoder */

long sealhip_context_using_batching(const sealhip_context *ctx, int32_t *using_batching)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(using_batching);
    *using_batching = ctx->engine->plain_prime >= 0 ? 1 : 0;
    return SEALHIP_S_OK;
}

long sealhip_batch_encode(sealhip_context *ctx, const uint64_t *values, size_t n_values, size_t count, uint64_t *plain)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(values);
    REQUIRE_PTR(plain);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        op_batch_encode(e, reinterpret_cast<const u64 *>(values), n_values, count, reinterpret_cast<u64 *>(plain));
    });
}

long sealhip_batch_decode(sealhip_context *ctx, const uint64_t *plain, size_t count, uint64_t *values)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(plain);
    REQUIRE_PTR(values);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        op_batch_decode(e, reinterpret_cast<const u64 *>(plain), count, reinterpret_cast<u64 *>(values));
    });
}

long sealhip_batch_encode_int64(sealhip_context *ctx, const int64_t *values, size_t n_values, size_t count, uint64_t *plain)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(values);
    REQUIRE_PTR(plain);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        op_batch_encode(e, reinterpret_cast<const u64 *>(values), n_values, count, reinterpret_cast<u64 *>(plain), true);
    });
}

long sealhip_batch_decode_int64(sealhip_context *ctx, const uint64_t *plain, size_t count, int64_t *values)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(plain);
    REQUIRE_PTR(values);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        op_batch_decode(e, reinterpret_cast<const u64 *>(plain), count, reinterpret_cast<u64 *>(values), true);
    });
}

long sealhip_ckks_encode(sealhip_context *ctx, uint32_t k, const double *values, size_t n_values, size_t count, double scale,
                         uint64_t *plain)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(plain);
    if (n_values)
        REQUIRE_PTR(values);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        op_ckks_encode(e, static_cast<int>(k), values, n_values, count, scale, reinterpret_cast<u64 *>(plain));
    });
}

// CKKSEncoder::encode_internal(double value, ...) (ckks.cpp:80-216): every slot holds `value`, i.e. the plaintext is the
// constant polynomial round(value * scale), whose NTT form is that constant in every position of every row. The three
// decomposition branches of the reference (<= 64 bits, <= 128 bits, multi-precision) are exact decompositions of one
// integer -- a 53-bit mantissa times a power of two -- so one exact path gives their residues.
long sealhip_ckks_encode_value(sealhip_context *ctx, uint32_t k, double value, double scale, size_t count, uint64_t *plain)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(plain);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (e.scheme != 2)
            throw std::invalid_argument("unsupported scheme"); // ckks.cpp:27-30
        const int total_bits = e.total_coeff_modulus_bit_count(static_cast<int>(k));
        if (scale <= 0 || (static_cast<int>(std::log2(scale)) >= total_bits))
            throw std::invalid_argument("scale out of bounds"); // :105-109
        value *= scale; // :112
        const int coeff_bit_count = static_cast<int>(std::log2(std::fabs(value))) + 2;
        if (coeff_bit_count >= total_bits)
            throw std::invalid_argument("encoded value is too large"); // :114-118
        double coeffd = std::round(value);
        const bool is_negative = std::signbit(coeffd);
        coeffd = std::fabs(coeffd);
        int exp2 = 0;
        const double frac = std::frexp(coeffd, &exp2);                      // coeffd = frac * 2^exp2, frac in [0.5, 1)
        const u64 mant = static_cast<u64>(std::ldexp(frac, 53));             // exact: 53-bit mantissa
        const int shift = exp2 - 53;                                          // coeffd = mant * 2^shift (shift may be < 0)
        u64 rows[kMaxModuli];
        for (uint32_t j = 0; j < k; j++)
        {
            const u64 q = e.key_moduli[j];
            u64 r;
            if (coeffd == 0)
                r = 0;
            else if (shift <= 0)
                r = (mant >> (-shift)) % q; // an integer: the low -shift bits of the mantissa are zero
            else
            {
                u64 pw = 1 % q, base = 2 % q; // 2^shift mod q
                for (int s2 = shift; s2; s2 >>= 1)
                {
                    if (s2 & 1)
                        pw = static_cast<u64>(static_cast<unsigned __int128>(pw) * base % q);
                    base = static_cast<u64>(static_cast<unsigned __int128>(base) * base % q);
                }
                r = static_cast<u64>(static_cast<unsigned __int128>(mant % q) * pw % q);
            }
            rows[j] = is_negative ? (r ? q - r : 0) : r; // negate_uint_mod, :137-140
        }
        check_launch(launch_fill_rows(e, reinterpret_cast<u64 *>(plain), rows, static_cast<int>(k), count), "ckks encode value");
    });
}

// CKKSEncoder::encode_internal(int64_t value, ...) (ckks.cpp:218-275): scale 1.0, residues exactly as the reference forms them
long sealhip_ckks_encode_int64(sealhip_context *ctx, uint32_t k, int64_t value, size_t count, uint64_t *plain)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(plain);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (e.scheme != 2)
            throw std::invalid_argument("unsupported scheme");
        const u64 mag = static_cast<u64>(value < 0 ? -(value + 1) : value) + (value < 0 ? 1u : 0u); // llabs without overflow
        const int bits = mag ? 64 - __builtin_clzll(mag) : 0;
        if (bits + 2 >= e.total_coeff_modulus_bit_count(static_cast<int>(k)))
            throw std::invalid_argument("encoded value is too large"); // :240-244
        u64 rows[kMaxModuli];
        for (uint32_t j = 0; j < k; j++)
        {
            const u64 q = e.key_moduli[j];
            u64 tmp = static_cast<u64>(value);
            if (value < 0)
                tmp += q; // :254-257 (wrapping, as written there)
            rows[j] = tmp % q;
        }
        check_launch(launch_fill_rows(e, reinterpret_cast<u64 *>(plain), rows, static_cast<int>(k), count), "ckks encode int64");
    });
}

long sealhip_ckks_decode(sealhip_context *ctx, uint32_t k, const uint64_t *plain, size_t count, double scale, double *values)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(plain);
    REQUIRE_PTR(values);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        op_ckks_decode(e, static_cast<int>(k), reinterpret_cast<const u64 *>(plain), count, scale, values);
    });
}

long sealhip_ciphertext_resize(sealhip_context *ctx, uint32_t k, const uint64_t *src, uint32_t src_size, uint64_t *dst,
                               uint32_t dst_size, size_t count)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(src);
    REQUIRE_PTR(dst);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if ((dst_size < 2 && dst_size != 0) || dst_size > 16 || src_size > 16)
            throw std::invalid_argument("invalid size"); // ciphertext.cpp:111-114
        if (count == 0 || dst_size == 0)
            return;
        const std::size_t poly = static_cast<std::size_t>(k) * e.n;
        const uint32_t keep = std::min(src_size, dst_size);
        u64 *d = reinterpret_cast<u64 *>(dst);
        if (dst_size > keep)
            SEALHIP_CHECK(hipMemsetAsync(d, 0, count * dst_size * poly * sizeof(u64), e.lane().stream));
        if (keep)
            check_launch(launch_copy_rows(e, reinterpret_cast<const u64 *>(src), src_size * poly, d, dst_size * poly, count,
                                          static_cast<int>(keep * k)),
                         "resize");
    });
}

/* ------------------------------------------------------------------ ciphertext wire format */

long sealhip_context_set_parms_id(sealhip_context *ctx, uint32_t k, const uint64_t parms_id[4])
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(parms_id);
    return guarded([&] { wire_set_parms_id(*ctx->engine, static_cast<int>(k), parms_id); });
}

long sealhip_ciphertext_peek(const void *bytes, size_t len, sealhip_ciphertext_info *info)
{
    REQUIRE_PTR(bytes);
    REQUIRE_PTR(info);
    return guarded([&] { wire_peek(bytes, len, info); });
}

long sealhip_ciphertext_load(sealhip_context *ctx, const void *bytes, size_t len, sealhip_ciphertext_info *info,
                             uint64_t *dst_device, size_t capacity_words)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(bytes);
    REQUIRE_PTR(info);
    REQUIRE_PTR(dst_device);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        wire_load(e, bytes, len, info, reinterpret_cast<u64 *>(dst_device), capacity_words);
    });
}

long sealhip_ciphertext_save_size(const sealhip_context *ctx, uint32_t size, uint32_t k, size_t *bytes)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(bytes);
    *bytes = wire_save_size(size, k, ctx->engine->n);
    return SEALHIP_S_OK;
}

long sealhip_ciphertext_save(sealhip_context *ctx, const sealhip_ciphertext_info *info, const uint64_t *src_device,
                             void *bytes, size_t capacity, size_t *written)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(info);
    REQUIRE_PTR(src_device);
    REQUIRE_PTR(bytes);
    REQUIRE_PTR(written);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        *written = wire_save(e, *info, reinterpret_cast<const u64 *>(src_device), bytes, capacity);
    });
}

long sealhip_kswitch_key_load_stream(sealhip_context *ctx, const void *bytes, size_t len, uint32_t index,
                                     sealhip_kswitch_key **key, uint64_t *n_slots)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(bytes);
    REQUIRE_PTR(key);
    *key = nullptr;
    return guarded([&] {
        Engine &e = device_engine(ctx);
        u64 *dev = nullptr;
        std::size_t words = 0;
        std::uint64_t dim1 = 0;
        const std::uint32_t digits = wire_load_kswitch_key(e, bytes, len, index, &dev, &words, &dim1);
        if (n_slots)
            *n_slots = dim1;
        if (!digits)
            return;
        auto k = std::make_unique<sealhip_kswitch_key>();
        k->key.n_digits = digits;
        k->key.words = words;
        k->key.d_data = dev;
        *key = k.release();
    });
}

long sealhip_debug_blake2xb(void *out, size_t outlen, const void *in, size_t inlen, const void *key, size_t keylen)
{
    REQUIRE_PTR(out);
    return guarded([&] {
        if (!blake2xb(out, outlen, in, inlen, key, keylen))
            throw std::invalid_argument("blake2xb: bad arguments");
    });
}

long sealhip_expand_seed_host(sealhip_context *ctx, uint32_t rows, const uint64_t seed[8], uint64_t *out_host)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(seed);
    REQUIRE_PTR(out_host);
    return guarded([&] {
        Engine &e = *ctx->engine; // host work: also on host-only contexts
        if (rows < 1 || static_cast<int>(rows) > e.n_key)
            throw std::invalid_argument("level k out of range");
        const std::vector<u64> c1 = wire_expand_seed(e, static_cast<int>(rows), reinterpret_cast<const unsigned char *>(seed));
        std::memcpy(out_host, c1.data(), c1.size() * sizeof(u64));
    });
}

long sealhip_kswitch_keys_save(sealhip_context *ctx, const sealhip_kswitch_key *const *keys, uint32_t n_slots, void *bytes,
                               size_t capacity, size_t *written)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(written);
    if (n_slots)
        REQUIRE_PTR(keys);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        std::vector<const KSwitchKey *> raw(n_slots, nullptr);
        for (uint32_t i = 0; i < n_slots; i++)
            raw[i] = keys[i] ? &keys[i]->key : nullptr;
        if (!bytes) // size query
        {
            *written = wire_kswitch_save_size(e, raw.data(), n_slots);
            return;
        }
        *written = wire_save_kswitch_keys(e, raw.data(), n_slots, bytes, capacity);
    });
}

long sealhip_is_data_valid_for(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size, size_t count,
                               uint8_t *valid)
{
    REQUIRE_PTR(ctx);
    REQUIRE_PTR(ct);
    REQUIRE_PTR(valid);
    return guarded([&] {
        Engine &e = device_engine(ctx);
        check_level(e, k);
        if (count == 0)
            return;
        RowMap map{};
        map.rows = static_cast<int>(k);
        for (uint32_t r = 0; r < k; r++)
            map.prime[r] = static_cast<unsigned short>(r);
        e.ws_reserve(e.lane().ws_floor + count * sizeof(unsigned) + 512);
        e.ws_reset();
        unsigned *flags = reinterpret_cast<unsigned *>(e.ws_alloc((count * sizeof(unsigned) + 7) / 8));
        SEALHIP_CHECK(hipMemsetAsync(flags, 0, count * sizeof(unsigned), e.lane().stream));
        check_launch(launch_out_of_range(e, reinterpret_cast<const u64 *>(ct), static_cast<std::size_t>(size) * k * e.n, count,
                                         map, flags),
                     "is_data_valid_for");
        std::vector<unsigned> host(count);
        SEALHIP_CHECK(hipMemcpyAsync(host.data(), flags, count * sizeof(unsigned), hipMemcpyDeviceToHost, e.lane().stream));
        e.sync_and_check();
        for (size_t i = 0; i < count; i++)
            valid[i] = host[i] ? 0 : 1;
    });
}

} // extern "C"
