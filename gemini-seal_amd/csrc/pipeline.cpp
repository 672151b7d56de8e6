// pipeline.cpp -- the Evaluator-level operations composed from the kernels: BFV/CKKS multiply
// (native/src/seal/evaluator.cpp:274-527), hybrid key switch (:2259-2368), modulus switch / rescale
// (:829-957) and apply_galois (:1841-1943), batched over independent ciphertexts. Large batches are
// processed in chunks sized to the workspace arena; every chunk is a fixed sequence of launches on
// one stream, so temporaries are reused in stream order without host synchronisation.
#include "engine.hpp"

#include <cmath>

#include <algorithm>
#include <cstdlib>

namespace sealhip
{
    namespace
    {
        std::size_t workspace_budget_bytes(const Engine &e)
        {
            // cap of a lane's temporaries arena (it grows on demand up to this): SEALHIP_WORKSPACE_MB, else a sixth of the
            // memory free on the device when the lane first needs it, between 2 and 48 GiB -- sized for 288 GB of HBM,
            // where larger chunks mean larger launches (bench: +3 % from 8 to 48 GiB). Fixed per lane (an operation that
            // parks scratch at the front of the arena relies on the nested operation seeing the same cap); a lane created
            // later sees what the earlier ones left, so the caps of all threads' lanes cannot add up to more than the device.
            Lane &l = e.lane();
            if (l.ws_budget)
                return l.ws_budget;
            static const std::size_t env_budget = [] {
                if (const char *env = std::getenv("SEALHIP_WORKSPACE_MB"))
                {
                    std::size_t mb = static_cast<std::size_t>(std::strtoull(env, nullptr, 10));
                    return (mb < 64 ? std::size_t(64) : mb) << 20;
                }
                return std::size_t(0);
            }();
            if (env_budget)
                return l.ws_budget = env_budget;
            std::size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
                return l.ws_budget = std::size_t(8192) << 20;
            const std::size_t lo = std::size_t(2) << 30, hi = std::size_t(48) << 30;
            return l.ws_budget = std::min(hi, std::max(lo, (free_b + l.ws_bytes) / 6));
        }

        std::size_t pad256(std::size_t words)
        {
            return ((words * sizeof(u64) + 255) & ~static_cast<std::size_t>(255));
        }

        // how many items fit the arena, given the padded byte need of one item (sum over its buffers)
        std::size_t plan_chunk(Engine &e, std::size_t count, std::size_t bytes_per_item, int n_buffers)
        {
            const std::size_t budget = workspace_budget_bytes(e);
            std::size_t chunk = budget / (bytes_per_item ? bytes_per_item : 1);
            chunk = std::max<std::size_t>(1, std::min(chunk, count));
            auto &log = e.lane().chunk_log;
            if (log.size() >= 64)
                log.erase(log.begin());
            log.emplace_back(count, chunk);
            e.ws_reserve(e.lane().ws_floor + chunk * bytes_per_item + static_cast<std::size_t>(n_buffers) * 256);
            return chunk;
        }

        void check(hipError_t err, const char *what)
        {
            if (err != hipSuccess)
                throw HipError(err, (std::string(what) + ": " + hipGetErrorString(err)).c_str());
        }

        RowMap skip_map(const RowMap &base, int keep_lo, int keep_hi)
        {
            RowMap m = base;
            for (int r = 0; r < m.rows; r++)
                if (r < keep_lo || r >= keep_hi)
                    m.prime[r] = kSkipRow;
            return m;
        }
    } // namespace

    // bytes of arena one item of op_switch_key needs (shared with op_apply_galois, which must size the arena
    // before it parks its own scratch at the front)
    static std::size_t switch_key_item_bytes(Engine &e, int k)
    {
        LevelTools &lt = e.level(k);
        const std::size_t N = e.n;
        const std::size_t rows = static_cast<std::size_t>(k + e.nsp), nd = static_cast<std::size_t>(lt.h_ks.nd);
        const bool ckks = e.scheme == 2;
        const bool strict_bfv = e.mode_strict && !ckks;
        const std::size_t w_coeff = (ckks || strict_bfv) ? static_cast<std::size_t>(k) * N : 0;
        return (w_coeff + nd * rows * N + 2 * rows * N + 2 * static_cast<std::size_t>(k) * N) * sizeof(u64);
    }

    namespace
    {
        // Transparency sink (devmath.hpp note_nonzero): the sink of the operation in flight, shifted to a chunk's first
        // item, and the scope around the ONE launch that stores the result's polynomials 1..
        unsigned *sink_at(Engine &e, std::size_t off)
        {
            Lane &l = e.lane();
            return l.tsink_cur ? l.tsink_cur + l.tsink_base + off : nullptr;
        }
        struct SinkArm
        {
            Lane &l;
            SinkArm(Engine &e, unsigned *flags) : l(e.lane())
            {
                l.tsink_arm = flags;
            }
            ~SinkArm()
            {
                l.tsink_arm = nullptr;
            }
        };
    } // namespace

    // ------------------------------------------------------------------------------------------
    // switch_key_inplace (evaluator.cpp:2259-2368)
    // ------------------------------------------------------------------------------------------
    void op_switch_key(Engine &e, int k, u64 *ct, std::size_t ct_stride, const u64 *target, std::size_t target_stride,
                       std::size_t count, const KSwitchKey &key, const u64 *c0_src, std::size_t c0_stride, const KsSplit *split)
    {
        if (k > e.k_first)
            throw std::invalid_argument("key switching needs a ciphertext level");
        LevelTools &lt = e.level(k);
        const KsDev &h = lt.h_ks;
        const bool finish = split && split->partial_sum;   // latency mode, after the all-reduce
        const bool partial = split && split->partial_out;  // latency mode, before it
        if (!finish && static_cast<int>(key.n_digits) < h.nd)
            throw std::invalid_argument("kswitch_keys is not valid for encryption parameters");
        const int dj0 = partial ? split->j0 : 0, dj1 = partial ? split->j1 : h.nd;
        if (dj0 < 0 || dj0 > dj1 || dj1 > h.nd)
            throw std::invalid_argument("digit range out of bounds");
        const std::size_t N = e.n;
        const int rows = k + e.nsp, nd = h.nd;
        const bool ckks = e.scheme == 2;
        const bool strict_bfv = e.mode_strict && !ckks;
        // per item words
        const std::size_t w_coeff = (ckks || strict_bfv) ? static_cast<std::size_t>(k) * N : 0;
        const std::size_t w_ext = static_cast<std::size_t>(nd) * rows * N;
        const std::size_t w_prod = 2ull * rows * N;
        const std::size_t w_temp = 2ull * k * N;
        const std::size_t per_item = switch_key_item_bytes(e, k);
        if (per_item != (w_coeff + w_ext + w_prod + w_temp) * sizeof(u64))
            throw std::logic_error("internal: arena accounting mismatch");
        const std::size_t chunk = plan_chunk(e, count, per_item, 4);
        const RowMap map_q = lt.map_q;
        const RowMap map_rows = lt.map_key;
        for (std::size_t off = 0; off < count; off += chunk)
        {
            const std::size_t m = std::min(chunk, count - off);
            e.ws_reset();
            u64 *coeff = w_coeff ? e.ws_alloc(w_coeff * m) : nullptr;
            u64 *ext = e.ws_alloc(w_ext * m);
            u64 *prod = e.ws_alloc(w_prod * m);
            u64 *temp = e.ws_alloc(w_temp * m);
            if (partial)
                prod = split->partial_out + off * w_prod;
            if (finish)
                prod = split->partial_sum + off * w_prod;
            const u64 *tg = finish ? nullptr : target + off * target_stride;
            u64 *ctp = partial ? nullptr : ct + off * ct_stride;
            const u64 *c0p = c0_src ? c0_src + off * c0_stride : nullptr;
            const std::size_t ext_item = static_cast<std::size_t>(rows) * N;
            const std::size_t ext_digit = ext_item * m; // digit-major

            if (finish)
                // the summed canonical partials (below ranks * p < 2^63) back to canonical residues: from here on every word
                // is what the unsplit inner product (:2341-2349) leaves
                check(launch_poly_op(e, PolyOp::Mod63, prod, nullptr, 0, prod, m * 2 * rows, map_rows), "mod(partial sum)");
            // Step 1 (:2302-2307): CKKS bundles go back to coefficient form (canonical inverse NTT)
            const u64 *src = tg;
            std::size_t src_stride = target_stride;
            if (ckks && !finish)
            {
                // (valid ciphertext rows are below p and the canonicalising top kernel follows: any representative will do)
                if (ntt_can_gather(e)) // the inverse kernel reads the target rows where they are
                    check(launch_intt_from(e, coeff, tg, target_stride, m * k, map_q, kNttCanonical | kNttAnyRep),
                          "intt(target)");
                else
                {
                    check(launch_copy_rows(e, tg, target_stride, coeff, static_cast<std::size_t>(k) * N, m, k), "copy");
                    check(launch_ntt(e, coeff, m * k, map_q, true, kNttCanonical), "intt(target)");
                }
                src = coeff;
                src_stride = static_cast<std::size_t>(k) * N;
            }
            // Step 2 (:2310) + 3a (:2322): mod-up of every bundle, then the lazy forward NTT of every row outside
            // the bundle. With one special prime the mod-up of a row is a copy or a Barrett-63 reduction of the
            // bundle's single row (multi_special_primes.cpp:99-108): the NTT kernel gathers and reduces it on load,
            // so the extended polynomial is never written in coefficient form.
            const bool gather = e.nsp == 1 && ntt_can_gather(e);
            // launch-wide reduction of the gathered words: none when no source prime exceeds a destination prime, one
            // conditional subtraction when every source prime is below twice every destination prime, else Barrett-63
            bool need_any = false, need_barrett = false;
            for (int a = 0; a < k; a++)
                for (int b = 0; b < rows; b++)
                {
                    const u64 ps = e.key_moduli[h.row_prime[a]], pd = e.key_moduli[h.row_prime[b]];
                    need_any = need_any || (a != b && ps > pd);
                    need_barrett = need_barrett || (a != b && ps >= 2 * pd);
                }
            int modup_mode = need_barrett ? 1 : (need_any ? 2 : 0);
            if (modup_mode == 2)
            {
                // A gathered word is then below 2p of its destination prime. The lazy forward transform takes such an input
                // as it is (the first operand of a butterfly is never reduced before the last layer, the second goes through
                // mulmodLazy, ntt.cpp:245-261): with p < 2^58 nothing can wrap (2p(log n + 1) < 2^64), the extended
                // polynomial never leaves this function, and the inner product reduces to the canonical residue either
                // way -- so the conditional subtraction of all loaded words (6 % of the kernel's vector instructions, and
                // the kernel runs at the package power cap) is dropped. Larger primes keep it.
                u64 pmax = 0;
                for (int r = 0; r < rows; r++)
                    pmax = std::max(pmax, e.key_moduli[h.row_prime[r]]);
                if (bounds::fwd_lazy_admits(pmax, e.logn)) // (inputs below 2p: the case the recurrence in ntt_bounds.hpp walks)
                    modup_mode = 0;
            }
            if (!gather && !finish)
                check(launch_ks_modup(e, lt.d_ks, h, src, src_stride, ext, ext_item, ext_digit, m, -1), "modup");
            for (int j = dj0; j < dj1 && !finish; j++)
            {
                RowMap mj = map_rows;
                const int r0 = j * e.nsp, r1 = std::min(r0 + e.nsp, k);
                for (int r = r0; r < r1; r++)
                    mj.prime[r] = kSkipRow;
                if (gather)
                {
                    NttSource ns{};
                    ns.base[0] = src;
                    ns.poly_stride[0] = src_stride;
                    ns.reduce_mode = modup_mode;
                    const u64 p_src = e.key_moduli[h.row_prime[j]];
                    for (int r = 0; r < rows; r++)
                    {
                        if (r == j)
                        {
                            ns.code[r] = kSkipRow;
                            continue;
                        }
                        const u64 p_dst = e.key_moduli[h.row_prime[r]];
                        ns.code[r] = static_cast<unsigned short>(j | (p_src <= p_dst ? 0 : kSrcReduce));
                    }
                    // (the inner product reduces canonically: any representative of the transformed rows will do)
                    check(launch_ntt_gather(e, ext + j * ext_digit, m * rows, mj, ns, kNttAnyRep | kNttApprox), "ntt(ext, gathered)");
                }
                else
                    check(launch_ntt(e, ext + j * ext_digit, m * rows, mj, false, kNttAnyRep), "ntt(ext)"); // (as above)
            }
            // in-bundle rows: the reference multiplies the target rows as they are (:2319-2320);
            // STRICT transforms the coefficient-form BFV rows first
            const u64 *inb = tg;
            std::size_t inb_stride = target_stride;
            if (strict_bfv && !finish)
            {
                if (ntt_can_gather(e))
                {
                    // the transform reads the target rows where they are (no copy pass); the inner product reduces canonically,
                    // so any representative of the transformed rows will do, as for the digit rows above
                    NttSource ns{};
                    ns.base[0] = tg;
                    ns.poly_stride[0] = target_stride;
                    for (int r = 0; r < k; r++)
                        ns.code[r] = static_cast<unsigned short>(r);
                    check(launch_ntt_gather(e, coeff, m * k, map_q, ns, kNttAnyRep | kNttApprox), "ntt(target, gathered)");
                }
                else
                {
                    check(launch_copy_rows(e, tg, target_stride, coeff, static_cast<std::size_t>(k) * N, m, k), "copy");
                    check(launch_ntt(e, coeff, m * k, map_q, false, 0), "ntt(target)");
                }
                inb = coeff;
                inb_stride = static_cast<std::size_t>(k) * N;
            }
            // Step 3b + 4 (:2326-2349): 128-bit inner product over the digits, reduced
            if (!finish)
                check(launch_ks_mac(e, lt.d_ks, h, inb, inb_stride, ext, ext_item, ext_digit, key.d_data, prod, w_prod, m, dj0, dj1),
                      "mac");
            if (partial)
                continue; // the reduced partial products leave here (all-reduce, then op_switch_key with partial_sum)
            if (!ckks)
            {
                // BFV: every row of both products goes back to coefficient form in one launch (the reference does the
                // special rows first, :2351-2355, and the others inside rescale_special_rns_inplace, :286-289 -- the
                // order of independent row transforms does not matter), then one fused mod-down kernel
                const bool defer = ntt_can_defer_top(e, k);
                // (ks_moddown_bfv reduces what it reads canonically: any representative below 2p will do)
                check(launch_ntt(e, prod, m * 2 * rows, map_rows, true, (defer ? kNttDeferTop : 0) | kNttAnyRep), "intt(prod)");
                SinkArm arm(e, sink_at(e, off)); // (component 1 of the m ciphertexts of this chunk)
                check(launch_ks_moddown_bfv(e, lt.d_ks, h, prod, ext_item, ctp, ct_stride, 2 * m, defer, c0p, c0_stride),
                      "moddown_bfv");
                continue;
            }
            const u64 p_special = (ckks && e.nsp == 1) ? e.key_moduli[h.row_prime[k]] : 0;
            // The gathered transform is handed the integer P - r < P instead of the residue (-(s mod P)) mod q_i: the same
            // word wherever P <= q_i; where q_i < P < 2 q_i it is an unreduced input below 2 q_i, which the lazy forward
            // transform takes as it is only while nothing can wrap (ntt_bounds.hpp section 2, inputs below 2p).
            bool fold_ok = ckks && e.nsp == 1 && ntt_can_gather(e) && exp_env("SEALHIP_KS_MODDOWN_UNFUSED") == nullptr;
            for (int r = 0; fold_ok && r < k; r++)
            {
                const u64 q = e.key_moduli[h.row_prime[r]];
                fold_ok = p_special <= q || (p_special < 2 * q && bounds::fwd_lazy_admits(q, e.logn));
            }
            const bool fold_pre = fold_ok;
            // the gathered transform below reads the special row as pairs (c, c + N/2): it can apply the top inverse layer
            const bool fold_top = fold_pre && ntt_can_defer_top(e, k);
            // ... and, in its floating-point form, finish the mod-down as it stores (reduce mode 7): temp is never written
            const bool fold_store = fold_top && ntt_can_fuse_moddown(e, k, p_special);
            // (:2351-2355) special rows back to coefficient form (lazy)
            // (the mod-down reduces the special rows with barrett_reduce_63 / a Shoup product: canonical either way)
            check(launch_ntt(e, prod, m * 2 * rows, skip_map(map_rows, k, rows), true, kNttAnyRep | (fold_top ? kNttDeferTop : 0)),
                  "intt(special)");
            // Step 5 (:2361): rescale_special_rns_inplace, then add into the ciphertext (:2363-2366)
            // Step 5 for CKKS with one special prime P: temp_i = (-(special mod P)) mod q_i, then its forward transform
            // (multi_special_primes.cpp:262-289). With the single-pass kernel the transform gathers the special row and forms
            // temp_i while it loads (reduce mode 4): no pass that writes the k rows of temp, none that reads them back.
            if (fold_pre)
            {
                NttSource ns{};
                ns.base[0] = prod;
                ns.poly_stride[0] = ext_item;
                ns.reduce_mode = fold_store ? 7 : (fold_top ? 5 : 4);
                if (fold_store)
                {
                    ns.md.inv_p = lt.d_ks->invP; // (addresses inside the device copy of KsDev; not dereferenced here)
                    ns.md.inv_p_shoup = lt.d_ks->invP_shoup;
                    ns.md.prod = prod;
                    ns.md.prod_stride = ext_item;
                    ns.md.ct = ctp;
                    ns.md.ct_stride = ct_stride;
                    ns.md.c0_src = c0p;
                    ns.md.c0_stride = c0_stride;
                    ns.md.tflags = sink_at(e, off);
                }
                ns.aux_p = p_special;
                ns.aux_cr1 = HostModulus(p_special).cr1;
                {
                    const HostNttTables &tb = e.tables[h.row_prime[k]];
                    ns.aux_top[0] = tb.inv_n;
                    ns.aux_top[1] = tb.inv_n_shoup;
                    ns.aux_top[2] = tb.inv_n_w;
                    ns.aux_top[3] = tb.inv_n_w_shoup;
                }
                for (int r = 0; r < k; r++)
                    ns.code[r] = static_cast<unsigned short>(k); // every row of temp reads the special row of its polynomial
                // (ks_moddown_post adds these rows to the q rows and reduces the sum canonically: any representative will do)
                check(launch_ntt_gather(e, temp, m * 2 * k, map_q, ns, kNttAnyRep), "ntt(temp, gathered from the special row)");
            }
            else
            {
                check(launch_ks_moddown_pre(e, lt.d_ks, h, prod, ext_item, temp, static_cast<std::size_t>(k) * N, 2 * m),
                      "moddown_pre");
                if (ckks)
                    check(launch_ntt(e, temp, m * 2 * k, map_q, false, kNttAnyRep), "ntt(temp)"); // (moddown_post reduces the sum)
                else
                    check(launch_ntt(e, prod, m * 2 * rows, skip_map(map_rows, 0, k), true, kNttAnyRep), "intt(prod)");
            }
            if (!fold_store)
            {
                SinkArm arm(e, sink_at(e, off));
                check(launch_ks_moddown_post(e, lt.d_ks, h, prod, ext_item, temp, static_cast<std::size_t>(k) * N, ctp,
                                             ct_stride, 2 * m, 1, c0p, c0_stride),
                      "moddown_post");
            }
        }
    }

    // modup_rns as a standalone operation (multi_special_primes.cpp:151-185)
    void op_modup(Engine &e, int k, int bundle, u64 *ext, std::size_t count)
    {
        LevelTools &lt = e.level(k);
        if (k > e.k_first || bundle < 0 || bundle >= lt.h_ks.nd)
            throw std::invalid_argument("modup_rns: src_bundle_index out of bound");
        const std::size_t stride = static_cast<std::size_t>(k + e.nsp) * e.n;
        check(launch_ks_modup(e, lt.d_ks, lt.h_ks, ext, stride, ext, stride, 0, count, bundle), "modup");
    }

    // rescale_special_rns_inplace as a standalone operation (multi_special_primes.cpp:237-304)
    void op_rescale_special_inplace(Engine &e, int k, u64 *poly, std::size_t count)
    {
        if (k > e.k_first)
            throw std::invalid_argument("rescale_special needs a ciphertext level");
        LevelTools &lt = e.level(k);
        const std::size_t N = e.n;
        const int rows = k + e.nsp;
        const std::size_t per_item = static_cast<std::size_t>(k) * N * sizeof(u64);
        const std::size_t chunk = plan_chunk(e, count, per_item, 1);
        for (std::size_t off = 0; off < count; off += chunk)
        {
            const std::size_t m = std::min(chunk, count - off);
            e.ws_reset();
            u64 *temp = e.ws_alloc(static_cast<std::size_t>(k) * N * m);
            u64 *p = poly + off * rows * N;
            check(launch_ks_moddown_pre(e, lt.d_ks, lt.h_ks, p, static_cast<std::size_t>(rows) * N, temp,
                                        static_cast<std::size_t>(k) * N, m),
                  "moddown_pre");
            if (e.scheme == 2)
                check(launch_ntt(e, temp, m * k, lt.map_q, false, 0), "ntt(temp)");
            else
                check(launch_ntt(e, p, m * rows, skip_map(lt.map_key, 0, k), true, 0), "intt(poly)");
            check(launch_ks_moddown_post(e, lt.d_ks, lt.h_ks, p, static_cast<std::size_t>(rows) * N, temp,
                                         static_cast<std::size_t>(k) * N, nullptr, 0, m, 0),
                  "moddown_post");
        }
    }

    // ------------------------------------------------------------------------------------------
    // bfv_multiply (evaluator.cpp:274-445)
    // ------------------------------------------------------------------------------------------
    // b == nullptr: bfv_square of a size-2 ciphertext (evaluator.cpp:560-702) -- the two polynomials of `a` are lifted and
    // transformed once (:604-634) and the tensor product is c_0 = x_0^2, c_1 = x_0 x_1 added to itself, c_2 = x_1^2
    // (:644-657); everything after it is bfv_multiply's tail.
    void op_bfv_multiply(Engine &e, int k, const u64 *a, int sa, const u64 *b, int sb, std::size_t count, u64 *out)
    {
        LevelTools &lt = e.level(k);
        const RnsDev &h = lt.h_rns;
        const std::size_t N = e.n;
        const bool sq = b == nullptr;
        if (sq && (sa != 2 || sb != 2))
            throw std::logic_error("op_bfv_multiply: the square path takes size-2 operands");
        const int nB = h.nB, kb = k + nB, sin = sq ? sa : sa + sb, dest = sa + sb - 1;
        const std::size_t w_x = static_cast<std::size_t>(sin) * kb * N;
        const std::size_t w_d = static_cast<std::size_t>(dest) * kb * N;
        const std::size_t chunk = plan_chunk(e, count, (w_x + w_d) * sizeof(u64), 2);
        const std::size_t poly_q = static_cast<std::size_t>(k) * N, poly_x = static_cast<std::size_t>(kb) * N;
        for (std::size_t off = 0; off < count; off += chunk)
        {
            const std::size_t m = std::min(chunk, count - off);
            e.ws_reset();
            u64 *X = e.ws_alloc(w_x * m);
            u64 *D = e.ws_alloc(w_d * m);
            // steps (1)-(3) (:335-353): lift to Bsk (fastbconv_m_tilde + sm_mrq) and one lazy NTT over all rows;
            // the q rows are gathered straight from the operands by the NTT kernel (no set_poly copy) when the
            // single-pass kernel is available
            const bool gather = ntt_can_gather(e) && sin * kb <= kMaxRows;
            // With the single-pass kernels and two size-2 operands the tensor product (step 4) is formed by the inverse
            // NTT while it loads its rows (no separate pass over 7 rows per prime; launch_intt_tensor)
            const bool defer = ntt_can_defer_top(e, k);
            bool fused_tensor = gather && defer && sa == 2 && sb == 2 && dest * kb <= kMaxRows;
            // its Montgomery reduction lands below 2p on operands below 4p for ciphertext primes under 2^59 (the exact forward
            // sequence), on operands below (2 + g) p -- what an approximate-quotient launch without kNttAnyRep stores --
            // for primes under 2^57 (ntt_bounds.hpp section 3: tensor_admits_apx); the Bsk rows are stored below 2p
            bool tensor_apx = true;
            for (int r = 0; r < k; r++)
            {
                fused_tensor = fused_tensor && bounds::tensor_admits_4p(e.key_moduli[r]);
                tensor_apx = tensor_apx && bounds::tensor_admits_apx(e.key_moduli[r]);
            }
            for (int j = 0; j < nB; j++)
                fused_tensor = fused_tensor && bounds::tensor_admits_2p(e.tables[lt.map_qbsk.prime[k + j]].p);
            // the lift applies the forward transform's top layer to the Bsk rows it writes (kNttTopDone below)
            bool lift_top = fused_tensor && bfv_lift_can_apply_top(e, h);
            if (lift_top && e.mode_strict)
            {
                RowMap bsk{};
                bsk.rows = nB;
                for (int j = 0; j < nB; j++)
                    bsk.prime[j] = lt.map_qbsk.prime[k + j];
                lift_top = ntt_strict_top_done_ok(e, bsk);
            }
            for (int s = 0; s < sin; s++)
            {
                const bool first = s < sa;
                const u64 *src = first ? a + off * sa * poly_q + s * poly_q : b + off * sb * poly_q + (s - sa) * poly_q;
                const std::size_t src_stride = (first ? sa : sb) * poly_q;
                u64 *dst = X + s * poly_x;
                if (!gather)
                    check(launch_copy_rows(e, src, src_stride, dst, w_x, m, k), "copy");
                check(launch_bfv_lift(e, lt.d_rns, h, src, src_stride, dst + poly_q, w_x, m, lift_top), "bfv_lift");
            }
            if (gather)
            {
                // one "polynomial" of the launch = all sin*(k+|Bsk|) rows of an item
                RowMap big{};
                NttSource ns{};
                big.rows = sin * kb;
                ns.base[0] = a + off * sa * poly_q;
                ns.base[1] = sq ? nullptr : b + off * sb * poly_q;
                ns.poly_stride[0] = sa * poly_q;
                ns.poly_stride[1] = sb * poly_q;
                for (int s = 0; s < sin; s++)
                    for (int r = 0; r < kb; r++)
                    {
                        big.prime[s * kb + r] = lt.map_qbsk.prime[r];
                        unsigned short code = kSkipRow; // Bsk rows: in place (written by bfv_lift)
                        if (r < k)
                            code = static_cast<unsigned short>((s < sa ? 0 : kSrcSecond) | ((s < sa ? s : s - sa) * k + r));
                        ns.code[s * kb + r] = code;
                    }
                // two launches over disjoint rows: the q rows only feed the tensor product, which reduces canonically, so
                // their last layer may skip its Barrett step (kNttAnyRep); the 60-bit Bsk rows wrap in the reference (F2)
                // and keep its exact sequence. (The fused tensor product multiplies the q rows without reducing them first
                // and needs them below 5p: there the last layer keeps its Barrett step.) kNttApprox: the approximate Shoup
                // quotient where every prime is below 2^58 (one multiplier instruction less per butterfly).
                RowMap mq = big, mb = big;
                for (int s = 0; s < sin; s++)
                    for (int r = 0; r < kb; r++)
                        (r < k ? mb : mq).prime[s * kb + r] = kSkipRow;
                check(launch_ntt_gather(e, X, m * sin * kb, mq, ns,
                                        fused_tensor ? (tensor_apx ? kNttApprox : 0) : (kNttAnyRep | kNttApprox)),
                      "ntt(X, gathered q rows)");
                // (fused tensor product: the wrapped Bsk words are brought below 2p as they are stored -- the residue class
                //  is all the dyadic product depends on)
                // (kNttAnyRep next to kNttReduceOut: "any representative below 2p will do". PARITY launches drop it on these
                //  60-bit primes and keep the reference's words; STRICT launches take the dense lazy schedule with it, launch_half)
                check(launch_ntt(e, X, m * sin * kb, mb, false,
                                 fused_tensor ? (kNttReduceOut | kNttAnyRep | (lift_top ? kNttTopDone : 0)) : 0),
                      "ntt(X, Bsk rows)");
            }
            else
                // (the tensor product reduces whatever it reads, polyarithsmallmod.cpp:63-117; the 60-bit Bsk rows keep the
                //  reference's wrapped words either way: launch_half / the pass kernels only use the freedom on small primes)
                check(launch_ntt(e, X, m * sin * kb, lt.map_qbsk, false, kNttAnyRep), "ntt(X)");
            if (fused_tensor)
            {
                // steps (4) + (5) (:376-424): D[I] = inverse NTT of sum_{i1+i2=I} X[i1] (.) X[2+i2], top layer deferred, every
                // word with the Montgomery factor 2^-64 that bfv_floor_sk's constants undo. q rows: lazy sums (any
                // representative, sparse schedule); Bsk rows: operands reduced on load, the dense lazy schedule
                RowMap mq{}, mb{};
                mq.rows = mb.rows = dest * kb;
                for (int I = 0; I < dest; I++)
                    for (int r = 0; r < kb; r++)
                    {
                        mq.prime[I * kb + r] = r < k ? lt.map_qbsk.prime[r] : kSkipRow;
                        mb.prime[I * kb + r] = r < k ? kSkipRow : lt.map_qbsk.prime[r];
                    }
                check(launch_intt_tensor(e, D, X, w_x, poly_x, kb, m * dest * kb, mq, kNttDeferTop | kNttAnyRep, sq),
                      "intt(tensor, q rows)");
                // (round 4: the Bsk rows may store any representative below 2p as well -- bfv_floor_sk2 takes u, v below 2p and
                //  canonicalises; the launcher then runs the DENSE lazy schedule on the 60-bit primes, ntt_bounds.hpp section 1)
                check(launch_intt_tensor(e, D, X, w_x, poly_x, kb, m * dest * kb, mb, kNttDeferTop | kNttAnyRep, sq),
                      "intt(tensor, Bsk rows)");
                for (int I = 0; I < dest; I++)
                {
                    SinkArm arm(e, I >= 1 ? sink_at(e, off) : nullptr); // polynomials 1.. of the product
                    check(launch_bfv_floor_sk(e, lt.d_rns, h, D + I * poly_x, w_d, out + off * dest * poly_q + I * poly_q,
                                              dest * poly_q, m, 2),
                          "floor_sk");
                }
                continue;
            }
            // step (4) (:376-420)
            // (square: both operands are the same two transformed polynomials; the kernel then forms x_0 x_1 once and adds it
            //  to itself, :650-651)
            check(launch_tensor_product(e, X, sa, w_x, sq ? X : X + sa * poly_x, sb, w_x, D, w_d, m, lt.map_qbsk, sq), "tensor");
            // step (5) (:423-424); with the single-pass kernels the top inverse layer and the canonicalisation are
            // applied by the consumer while it loads (saves one read+write pass over D)
            if (defer)
            {
                // two launches over disjoint rows: the q rows may store any representative (bfv_floor_sk canonicalises
                // while it applies the deferred top layer), which lets the kernel drop most conditional subtractions
                // (sparse lazy schedule); so may the 60-bit Bsk rows, which take the dense schedule (round 4)
                RowMap mq = lt.map_qbsk, mb = lt.map_qbsk;
                for (int r = 0; r < kb; r++)
                    (r < k ? mb : mq).prime[r] = kSkipRow;
                check(launch_ntt(e, D, m * dest * kb, mq, true, kNttDeferTop | kNttAnyRep), "intt(D, q rows)");
                check(launch_ntt(e, D, m * dest * kb, mb, true, kNttDeferTop | kNttAnyRep), "intt(D, Bsk rows)");
            }
            else
                check(launch_ntt(e, D, m * dest * kb, lt.map_qbsk, true, kNttCanonical), "intt(D)");
            // steps (6)-(8) (:427-444)
            for (int I = 0; I < dest; I++)
            {
                SinkArm arm(e, I >= 1 ? sink_at(e, off) : nullptr);
                check(launch_bfv_floor_sk(e, lt.d_rns, h, D + I * poly_x, w_d, out + off * dest * poly_q + I * poly_q,
                                          dest * poly_q, m, defer ? 1 : 0),
                      "floor_sk");
            }
        }
    }

    void op_bfv_square(Engine &e, int k, const u64 *a, int sa, std::size_t count, u64 *out)
    {
        if (sa != 2)
            return op_bfv_multiply(e, k, a, sa, a, sa, count, out); // evaluator.cpp:579-583
        op_bfv_multiply(e, k, a, 2, nullptr, 2, count, out);
    }

    // ckks_square (evaluator.cpp:704-770) is its own path, like bfv_square: a size-2 operand goes through the tensor
    // kernel's square form -- c_0 = x_0^2, c_1 = x_0 x_1 added to itself (:752-758), c_2 = x_1^2: two polynomials read and
    // three written (five row passes per prime instead of seven), three products per coefficient instead of four; other
    // sizes go through ckks_multiply like the reference (:720-724). Parity: ref_ckks_square (oracle/sealref.c).
    void op_ckks_square(Engine &e, int k, const u64 *a, int sa, std::size_t count, u64 *out)
    {
        if (sa != 2)
            return op_ckks_multiply(e, k, a, sa, a, sa, count, out);
        LevelTools &lt = e.level(k);
        const std::size_t poly = static_cast<std::size_t>(k) * e.n;
        SinkArm arm(e, sink_at(e, 0));
        check(launch_tensor_product(e, a, 2, 2 * poly, a, 2, 2 * poly, out, 3 * poly, count, lt.map_q, true), "tensor (square)");
    }

    // ckks_multiply (evaluator.cpp:447-527)
    void op_ckks_multiply(Engine &e, int k, const u64 *a, int sa, const u64 *b, int sb, std::size_t count, u64 *out)
    {
        LevelTools &lt = e.level(k);
        const std::size_t poly = static_cast<std::size_t>(k) * e.n;
        SinkArm arm(e, sink_at(e, 0));
        check(launch_tensor_product(e, a, sa, sa * poly, b, sb, sb * poly, out, (sa + sb - 1) * poly, count, lt.map_q),
              "tensor");
    }

    // ------------------------------------------------------------------------------------------
    // mod_switch_scale_to_next (evaluator.cpp:829-892): BFV mod_switch_to_next / CKKS rescale_to_next
    // ------------------------------------------------------------------------------------------
    namespace
    {
        void mod_switch_polys(Engine &e, int k, const u64 *ct, std::size_t in_stride, u64 *out, std::size_t out_stride,
                              std::size_t npolys);
    }
    // in_item_stride (words, 0 = the ciphertexts are back to back): distance between consecutive ciphertexts of `ct` when they
    // sit in a wider container -- the size-2 result of relinearize inside its size-3 product (the reference's objects are
    // separate buffers; a contiguous batch needs the stride). Each component is then one strided pass.
    void op_mod_switch_scale(Engine &e, int k, const u64 *ct, int size, std::size_t count, u64 *out, std::size_t in_item_stride)
    {
        if (k < 2)
            throw std::invalid_argument("end of modulus switching chain reached"); // evaluator.cpp:1005-1008
        const std::size_t N = e.n;
        const std::size_t in_poly = static_cast<std::size_t>(k) * N, out_poly = static_cast<std::size_t>(k - 1) * N;
        if (in_item_stride == 0 || in_item_stride == size * in_poly)
            return mod_switch_polys(e, k, ct, in_poly, out, out_poly, count * size);
        if (in_item_stride < size * in_poly)
            throw std::invalid_argument("item stride smaller than one ciphertext");
        for (int comp = 0; comp < size; comp++)
            mod_switch_polys(e, k, ct + comp * in_poly, in_item_stride, out + comp * out_poly, size * out_poly, count);
    }

    namespace
    {
    void mod_switch_polys(Engine &e, int k, const u64 *ct, std::size_t in_stride, u64 *out, std::size_t out_stride,
                          std::size_t npolys)
    {
        LevelTools &lt = e.level(k);
        const std::size_t N = e.n;
        const std::size_t temp_stride = static_cast<std::size_t>(k - 1) * N;
        if (e.scheme == 1)
        {
            check(launch_divround_bfv(e, lt.d_rns, lt.h_rns, ct, in_stride, out, out_stride, npolys, k - 1), "divround");
            return;
        }
        // CKKS (rns.cpp:777-851), without the reference's full copy of the ciphertext: only the last row is
        // duplicated because only it is modified
        const std::size_t per_poly = (N + static_cast<std::size_t>(k - 1) * N) * sizeof(u64);
        const std::size_t chunk = plan_chunk(e, npolys, per_poly, 2);
        const RowMap map_low = e.level_host(k - 1).map_q;
        for (std::size_t off = 0; off < npolys; off += chunk)
        {
            const std::size_t m = std::min(chunk, npolys - off);
            e.ws_reset();
            u64 *last = e.ws_alloc(N * m);
            u64 *temp = e.ws_alloc(temp_stride * m);
            check(launch_copy_rows(e, ct + off * in_stride + static_cast<std::size_t>(k - 1) * N, in_stride, last, N, m, 1),
                  "copy(last)");
            RowMap one{};
            one.rows = 1;
            one.prime[0] = static_cast<unsigned short>(k - 1);
            check(launch_ntt(e, last, m, one, true, kNttCanonical), "intt(last)");
            check(launch_rescale_pre(e, lt.d_rns, lt.h_rns, last, N, temp, temp_stride, m), "rescale_pre");
            check(launch_ntt(e, temp, m * (k - 1), map_low, false, 0), "ntt(temp)");
            check(launch_rescale_post(e, lt.d_rns, lt.h_rns, ct + off * in_stride, in_stride, temp, temp_stride,
                                      out + off * out_stride, out_stride, m),
                  "rescale_post");
        }
    }
    } // namespace

    // divide_and_round_q_last_ntt_inplace (rns.cpp:777-851), in place like the reference (last row clobbered)
    void op_divround_ntt_inplace(Engine &e, int k, u64 *data, std::size_t count)
    {
        if (k < 2)
            throw std::invalid_argument("divide_and_round_q_last needs at least two primes");
        LevelTools &lt = e.level(k);
        const std::size_t N = e.n;
        const std::size_t stride = static_cast<std::size_t>(k) * N, tstride = static_cast<std::size_t>(k - 1) * N;
        const std::size_t chunk = plan_chunk(e, count, tstride * sizeof(u64), 1);
        const RowMap map_low = e.level_host(k - 1).map_q;
        for (std::size_t off = 0; off < count; off += chunk)
        {
            const std::size_t m = std::min(chunk, count - off);
            e.ws_reset();
            u64 *temp = e.ws_alloc(tstride * m);
            u64 *p = data + off * stride;
            check(launch_ntt(e, p, m * k, skip_map(lt.map_q, k - 1, k), true, kNttCanonical), "intt(last)");
            check(launch_rescale_pre(e, lt.d_rns, lt.h_rns, p + static_cast<std::size_t>(k - 1) * N, stride, temp,
                                     tstride, m),
                  "rescale_pre");
            check(launch_ntt(e, temp, m * (k - 1), map_low, false, 0), "ntt(temp)");
            check(launch_rescale_post(e, lt.d_rns, lt.h_rns, p, stride, temp, tstride, p, stride, m), "rescale_post");
        }
    }

    // ------------------------------------------------------------------------------------------
    // apply_galois_inplace (evaluator.cpp:1841-1943)
    // ------------------------------------------------------------------------------------------
    void op_apply_galois(Engine &e, int k, u64 *ct, std::size_t count, std::uint32_t elt, const KSwitchKey &key)
    {
        const std::uint64_t m2 = static_cast<std::uint64_t>(e.n) * 2;
        if (!(elt & 1) || elt >= m2)
            throw std::invalid_argument("Galois element is not valid"); // :1880-1883
        LevelTools &lt = e.level(k);
        const std::size_t N = e.n, poly = static_cast<std::size_t>(k) * N;
        const std::uint32_t *table = e.scheme == 2 ? e.galois_table(elt) : nullptr;
        // The Galois image of both components (2 polys per item) must survive op_switch_key, which re-plans the
        // arena: it is carved from the FRONT of the arena (ws_floor) for the duration of this operation.
        const std::size_t per_item = 2 * poly * sizeof(u64);
        const std::size_t items = std::max<std::size_t>(1, std::min<std::size_t>(count, (1024ull << 20) / per_item));
        const std::size_t scratch_bytes = (items * per_item + 255) & ~static_cast<std::size_t>(255);
        // the arena must hold the scratch plus at least one item of the key switch; growing it may move it
        e.ws_reserve(scratch_bytes + 256);
        struct FloorGuard
        {
            Engine &e;
            std::size_t saved;
            ~FloorGuard()
            {
                e.lane().ws_floor = saved;
            }
        } guard{ e, e.lane().ws_floor };
        e.lane().ws_floor = guard.saved + scratch_bytes;
        for (std::size_t off = 0; off < count; off += items)
        {
            const std::size_t m = std::min(items, count - off);
            u64 *c = ct + off * 2 * poly;
            // (re-derive the pointer every iteration: a nested ws_reserve may have re-allocated the arena, but only
            //  while no scratch contents are live, i.e. before the first launch of this iteration)
            {
                // size the arena now exactly as the nested op_switch_key will ask for, so that it cannot move
                // while the scratch is live
                const std::size_t ks_item = switch_key_item_bytes(e, k);
                const std::size_t ks_chunk = std::max<std::size_t>(1, std::min(workspace_budget_bytes(e) / ks_item, m));
                e.ws_reserve(e.lane().ws_floor + ks_chunk * ks_item + 4 * 256);
            }
            u64 *scratch = reinterpret_cast<u64 *>(static_cast<char *>(e.lane().ws) + guard.saved);
            check(launch_galois(e, c, scratch, m * 2 * k, lt.map_q, elt, table), "galois");
            // The reference copies galois(c0) back (:1903 / :1917), zeroes c1 (:1928) and lets the key switch add its two
            // polynomials into that ciphertext (:1934-1935). Here the key switch's last kernel writes (galois(c0) + r0, r1)
            // directly: same sums, no copy pass and no fill pass over the ciphertext.
            e.lane().tsink_base = off; // (the nested chunk loop counts its items from this chunk's first ciphertext)
            struct BaseReset
            {
                Lane &l;
                ~BaseReset()
                {
                    l.tsink_base = 0;
                }
            } base_reset{ e.lane() };
            op_switch_key(e, k, c, 2 * poly, scratch + poly, 2 * poly, m, key, scratch, 2 * poly);
        }
    }
    // multiply_plain_normal (evaluator.cpp:1475-1603) for parameters with fast plain lift (every q_i > t): lift the
    // plaintext into the RNS base, canonical NTT, then per ciphertext polynomial lazy NTT -> dyadic product ->
    // canonical inverse NTT, in place. The monomial shortcut (:1516-1553) computes the same negacyclic product
    // exactly, so its canonical residues are identical to the generic path's; one path serves both.
    void op_multiply_plain(Engine &e, int k, u64 *ct, int size, std::size_t count, const u64 *plain,
                           std::size_t plain_stride)
    {
        if (e.scheme != 1)
            throw std::logic_error("unsupported operation for scheme type");
        const RowMap map_q = e.level_host(k).map_q;
        for (int r = 0; r < k; r++)
            if (e.key_moduli[r] <= e.t)
                throw std::logic_error("multiply_plain: parameters without fast plain lift are not supported");
        const std::size_t N = e.n;
        const std::size_t nplains = plain_stride ? count : 1;
        const std::size_t bytes = nplains * k * N * sizeof(u64);
        e.ws_reserve(e.lane().ws_floor + bytes + 256);
        e.ws_reset();
        u64 *temp = e.ws_alloc(nplains * k * N);
        check(launch_plain_lift(e, plain, plain_stride, temp, nplains, map_q, e.t), "plain_lift");
        check(launch_ntt(e, temp, nplains * k, map_q, false, kNttCanonical), "ntt(plain)");
        check(launch_ntt(e, ct, count * size * k, map_q, false, kNttAnyRep), "ntt(ct)"); // (the dyadic product reduces)
        check(launch_ct_linear(e, CtLinearOp::MulPlain, ct, size, temp, 0, plain_stride ? static_cast<std::size_t>(k) * N : 0,
                               ct, count, map_q),
              "dyadic(plain)");
        check(launch_ntt(e, ct, count * size * k, map_q, true, kNttCanonical), "intt(ct)");
    }
    // Decryptor::dot_product_ct_sk_array (decryptor.cpp:218-265): out[count][k][N] = c_0 + sum_{i>=1} c_i * s^i in the
    // form of the ciphertext. sk_powers = (size-1) polynomials s, s^2, ... in NTT form with key-level row stride.
    void op_dot_product_ct_sk(Engine &e, int k, const u64 *ct, int size, std::size_t count, const u64 *sk_powers,
                              bool is_ntt_form, u64 *out)
    {
        const RowMap map_q = e.level_host(k).map_q;
        const std::size_t N = e.n, poly = static_cast<std::size_t>(k) * N;
        const std::size_t sk_stride = static_cast<std::size_t>(e.n_key) * N;
        const std::size_t item = static_cast<std::size_t>(size) * poly;
        if (is_ntt_form || size == 1)
        {
            check(launch_dot_sk(e, ct, size, item, sk_powers, sk_stride, out, count, map_q, 1), "dot_sk");
            return;
        }
        // coefficient form: copies of c_1.. go to (lazy) NTT form (:241-244), the sum comes back canonical (:258-262),
        // then + c_0 (:265)
        const std::size_t tail = static_cast<std::size_t>(size - 1) * poly;
        const std::size_t chunk = plan_chunk(e, count, tail * sizeof(u64), 1);
        for (std::size_t off = 0; off < count; off += chunk)
        {
            const std::size_t m = std::min(chunk, count - off);
            e.ws_reset();
            u64 *copy = e.ws_alloc(tail * m);
            check(launch_copy_rows(e, ct + off * item + poly, item, copy, tail, m, (size - 1) * k), "copy(c1..)");
            check(launch_ntt(e, copy, m * (size - 1) * k, map_q, false, kNttAnyRep), "ntt(c1..)"); // (the dot product reduces)
            // the kernel indexes polynomials 1.. of an item: hand it a base one polynomial before the copies
            check(launch_dot_sk(e, copy - poly, size, tail, sk_powers, sk_stride, out + off * poly, m, map_q, 0), "dot_sk");
            check(launch_ntt(e, out + off * poly, m * k, map_q, true, kNttCanonical), "intt(dot)");
            check(launch_dot_sk(e, ct + off * item, 1, item, sk_powers, sk_stride, out + off * poly, m, map_q, 2), "add c0");
        }
    }
    // ---------------------------------------------------------------- encrypt-side arithmetic
    namespace
    {
        // rows 0..rows-1 of the key primes for each of `polys` polynomials of an item; `only` >= 0 keeps that polynomial
        RowMap ct_row_map(int rows, int polys, int only)
        {
            if (rows * polys > kMaxRows)
                throw std::invalid_argument("too many rows");
            RowMap m{};
            m.rows = rows * polys;
            for (int j = 0; j < polys; j++)
                for (int r = 0; r < rows; r++)
                    m.prime[j * rows + r] = (only < 0 || only == j) ? static_cast<unsigned short>(r) : kSkipRow;
            return m;
        }
    } // namespace

    void op_encrypt_zero_symmetric(Engine &e, int rows, bool is_ntt_form, const u64 *a_ntt, const std::int32_t *noise,
                                   const u64 *sk_ntt, std::size_t count, u64 *ct)
    {
        const std::size_t N = e.n, poly = static_cast<std::size_t>(rows) * N;
        u64 *c1 = ct + poly;
        if (c1 != a_ntt) // c_1 = a, sampled directly in NTT form (rlwe.cpp:245-249)
            check(launch_copy_rows(e, a_ntt, poly, c1, 2 * poly, count, rows), "copy(a)");
        RlweArgs a{};
        a.ct = ct;
        a.ct_item_stride = 2 * poly;
        a.ct_poly_stride = poly;
        a.polys = 1;
        a.rows = rows;
        a.x = c1;
        a.x_item_stride = 2 * poly;
        a.y = sk_ntt;
        a.e = noise;
        a.e_item_stride = N;
        a.negate = 1;
        if (is_ntt_form)
        {
            // noise to NTT form in the c_0 slot, then c_0 = -(noise + a*s)  (rlwe.cpp:266-284)
            check(launch_rlwe_stage(e, 0, a, count), "lift(e)");
            check(launch_ntt(e, ct, count * 2 * rows, ct_row_map(rows, 2, 0), false, kNttCanonical), "ntt(e)");
            check(launch_rlwe_stage(e, 2, a, count), "c0");
        }
        else
        {
            // c_0 = a*s back to coefficient form, c_0 = -(noise + c_0), and c_1 to coefficient form (:286-293)
            check(launch_rlwe_stage(e, 1, a, count), "a*s");
            check(launch_ntt(e, ct, count * 2 * rows, ct_row_map(rows, 2, -1), true, kNttCanonical), "intt(c0, c1)");
            check(launch_rlwe_stage(e, 3, a, count), "c0");
        }
    }

    void op_encrypt_zero_asymmetric(Engine &e, int rows, bool is_ntt_form, const u64 *pk, const std::int32_t *u,
                                    const std::int32_t *noise, std::size_t count, u64 *ct)
    {
        const std::size_t N = e.n, poly = static_cast<std::size_t>(rows) * N;
        const std::size_t chunk = plan_chunk(e, count, poly * sizeof(u64), 1);
        for (std::size_t off = 0; off < count; off += chunk)
        {
            const std::size_t m = std::min(chunk, count - off);
            e.ws_reset();
            u64 *u_ntt = e.ws_alloc(poly * m);
            RlweArgs a{};
            // u to RNS + NTT form (rlwe.cpp:161-170)
            a.ct = u_ntt;
            a.ct_item_stride = poly;
            a.polys = 1;
            a.rows = rows;
            a.e = u + off * N;
            a.e_item_stride = N;
            check(launch_rlwe_stage(e, 0, a, m), "lift(u)");
            check(launch_ntt(e, u_ntt, m * rows, ct_row_map(rows, 1, -1), false, kNttCanonical), "ntt(u)");
            // c_j = u * pk_j + e_j  (:171-201)
            a = RlweArgs{};
            a.ct = ct + off * 2 * poly;
            a.ct_item_stride = 2 * poly;
            a.ct_poly_stride = poly;
            a.polys = 2;
            a.rows = rows;
            a.x = u_ntt;
            a.x_item_stride = poly;
            a.x_poly_stride = 0;
            a.y = pk;
            a.y_poly_stride = poly;
            a.e = noise + off * 2 * N;
            a.e_item_stride = 2 * N;
            a.e_poly_stride = N;
            if (is_ntt_form)
            {
                check(launch_rlwe_stage(e, 0, a, m), "lift(e)");
                check(launch_ntt(e, a.ct, m * 2 * rows, ct_row_map(rows, 2, -1), false, kNttCanonical), "ntt(e)");
                check(launch_rlwe_stage(e, 2, a, m), "c = e + u*pk");
            }
            else
            {
                check(launch_rlwe_stage(e, 1, a, m), "u*pk");
                check(launch_ntt(e, a.ct, m * 2 * rows, ct_row_map(rows, 2, -1), true, kNttCanonical), "intt(c)");
                check(launch_rlwe_stage(e, 3, a, m), "c += e");
            }
        }
    }

    void op_scaling_variant(Engine &e, int k, const u64 *plain, std::size_t plain_item_stride, u64 *ct,
                            std::size_t ct_item_stride, std::size_t count, bool sub)
    {
        if (e.scheme != 1)
            throw std::invalid_argument("unsupported scheme");
        ScalingArgs a{};
        a.plain = plain;
        a.plain_item_stride = plain_item_stride;
        a.c0 = ct;
        a.c0_item_stride = ct_item_stride;
        a.k = k;
        a.sub = sub ? 1 : 0;
        a.t = e.t;
        HostModulus tm(e.t);
        a.t_cr0 = tm.cr0;
        a.t_cr1 = tm.cr1;
        a.threshold = (e.t + 1) >> 1; // plain_upper_half_threshold, context.cpp:317
        // coeff_div_plain_modulus = floor(q / t) in RNS form and q mod t (context.cpp:303-321)
        std::vector<u64> q(static_cast<std::size_t>(k), 0), quot(static_cast<std::size_t>(k), 0);
        q[0] = 1;
        for (int i = 0; i < k; i++)
        {
            u128 carry = 0;
            for (int l = 0; l < k; l++)
            {
                const u128 v = static_cast<u128>(q[l]) * e.key_moduli[i] + carry;
                q[l] = static_cast<u64>(v);
                carry = v >> 64;
            }
        }
        u128 rem = 0;
        for (int l = k; l-- > 0;)
        {
            const u128 cur = (rem << 64) | q[l];
            quot[l] = static_cast<u64>(cur / e.t);
            rem = cur % e.t;
        }
        a.q_mod_t = static_cast<u64>(rem);
        for (int j = 0; j < k; j++)
        {
            u128 r = 0;
            for (int l = k; l-- > 0;)
                r = ((r << 64) | quot[l]) % e.key_moduli[j];
            a.div[j] = static_cast<u64>(r);
        }
        check(launch_scaling_variant(e, a, count), "scaling_variant");
    }

    // ---------------------------------------------------------------- BatchEncoder
    namespace
    {
        RowMap plain_row_map(const Engine &e)
        {
            if (e.plain_prime < 0)
                throw std::invalid_argument("encryption parameters are not valid for batching"); // batchencoder.cpp:35-38
            RowMap m{};
            m.rows = 1;
            m.prime[0] = static_cast<unsigned short>(e.plain_prime);
            return m;
        }
    } // namespace

    void op_batch_encode(Engine &e, const u64 *values, std::size_t nvalues, std::size_t count, u64 *plain, bool is_signed)
    {
        const RowMap map = plain_row_map(e);
        if (nvalues > e.n)
            throw std::logic_error("values_matrix size is too large"); // batchencoder.cpp:119-122
        check(launch_batch_permute(e, true, values, nvalues, nvalues, plain, e.batch_map(), count, is_signed ? e.t : 0),
              "batch scatter");
        check(launch_ntt(e, plain, count, map, true, kNttCanonical), "intt(plain)");
    }

    void op_batch_decode(Engine &e, const u64 *plain, std::size_t count, u64 *values, bool is_signed)
    {
        const RowMap map = plain_row_map(e);
        const std::size_t N = e.n;
        const std::size_t chunk = plan_chunk(e, count, N * sizeof(u64), 1);
        for (std::size_t off = 0; off < count; off += chunk)
        {
            const std::size_t m = std::min(chunk, count - off);
            e.ws_reset();
            u64 *tmp = e.ws_alloc(N * m);
            check(launch_copy_rows(e, plain + off * N, N, tmp, N, m, 1), "copy(plain)");
            check(launch_ntt(e, tmp, m, map, false, kNttCanonical), "ntt(plain)");
            check(launch_batch_permute(e, false, tmp, N, N, values + off * N, e.batch_map(), m, is_signed ? e.t : 0), "batch gather");
        }
    }
    // ---------------------------------------------------------------- CKKSEncoder
    void op_ckks_encode(Engine &e, int k, const double *values, std::size_t n_values, std::size_t count, double scale,
                        u64 *plain)
    {
        if (e.scheme != 2)
            throw std::invalid_argument("unsupported scheme"); // ckks.cpp:27-30
        if (n_values > e.n / 2)
            throw std::invalid_argument("values_size is too large"); // ckks.h:419-422
        const int total_bits = e.total_coeff_modulus_bit_count(k);
        if (scale <= 0 || (static_cast<int>(std::log2(scale)) + 1 >= total_bits))
            throw std::invalid_argument("scale out of bounds"); // :440-444
        e.ckks_tables();
        const RowMap map_q = e.level_host(k).map_q;
        const std::size_t N = e.n;
        double n_inv = 1.0 / static_cast<double>(N); // :484-487
        n_inv *= scale;
        const std::size_t chunk = plan_chunk(e, count, N * 2 * sizeof(double), 2);
        int h_max = 1;
        for (std::size_t off = 0; off < count; off += chunk)
        {
            const std::size_t m = std::min(chunk, count - off);
            e.ws_reset();
            int *d_max = reinterpret_cast<int *>(e.ws_alloc(1));
            double *cv = reinterpret_cast<double *>(e.ws_alloc(2 * N * m));
            SEALHIP_CHECK(hipMemsetAsync(d_max, 0, sizeof(int), e.lane().stream));
            check(launch_ckks_encode_front(e, values + off * n_values * 2, n_values, m, n_inv, cv,
                                           plain + off * static_cast<std::size_t>(k) * N, k, e.d_ckks_map, e.d_ckks_inv_roots, d_max),
                  "ckks encode");
            int got = 0;
            SEALHIP_CHECK(hipMemcpyAsync(&got, d_max, sizeof(int), hipMemcpyDeviceToHost, e.lane().stream));
            SEALHIP_CHECK(hipStreamSynchronize(e.lane().stream));
            h_max = std::max(h_max, got);
        }
        if (h_max >= total_bits)
            throw std::invalid_argument("encoded values are too large"); // :501-504
        check(launch_ntt(e, plain, count * k, map_q, false, kNttCanonical), "ntt(plain)"); // :609-613
    }

    void op_ckks_decode(Engine &e, int k, const u64 *plain, std::size_t count, double scale, double *values)
    {
        if (e.scheme != 2)
            throw std::invalid_argument("unsupported scheme");
        if (scale <= 0 || (static_cast<int>(std::log2(scale)) >= e.total_coeff_modulus_bit_count(k)))
            throw std::invalid_argument("scale out of bounds"); // ckks.h:651-656
        e.ckks_tables();
        const CkksDecodeDev *consts = e.ckks_decode_consts(k);
        const RowMap map_q = e.level_host(k).map_q;
        const std::size_t N = e.n, poly = static_cast<std::size_t>(k) * N;
        const double inv_scale = 1.0 / scale; // :668
        const std::size_t chunk = plan_chunk(e, count, poly * sizeof(u64) + N * 2 * sizeof(double), 2);
        for (std::size_t off = 0; off < count; off += chunk)
        {
            const std::size_t m = std::min(chunk, count - off);
            e.ws_reset();
            u64 *copy = e.ws_alloc(poly * m);
            double *res = reinterpret_cast<double *>(e.ws_alloc(2 * N * m));
            if (ntt_can_gather(e)) // the single-pass inverse kernel reads the plaintext rows where they are
                check(launch_intt_from(e, copy, plain + off * poly, poly, m * k, map_q, kNttCanonical), "intt(plain)");
            else
            {
                check(launch_copy_rows(e, plain + off * poly, poly, copy, poly, m, k), "copy(plain)");
                check(launch_ntt(e, copy, m * k, map_q, true, kNttCanonical), "intt(plain)"); // :674-678
            }
            check(launch_ckks_decode_back(e, copy, consts, k, m, inv_scale, res, values + off * N, e.d_ckks_map, e.d_ckks_roots),
                  "ckks decode");
        }
    }
} // namespace sealhip
