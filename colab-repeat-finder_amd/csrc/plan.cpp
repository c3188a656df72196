// plan.cpp -- the work planner of the fused kernel: which wave runs which motif sizes, and how much LDS a workgroup takes.
// Pure host code without HIP headers: part of libprf.so, and built a second time with g++ -fsanitize=address,undefined
// (make asan -> libprf_host_asan.so) for the CPU suite (tests/test_asan_host.py).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "prf_plan.h"

using namespace prf_layout;
static constexpr int MAX_WAVES = PRF_VMAX_WAVES;

// ---------------------------------------------------------------------------------------------------
// Work plan: which wave runs which motif sizes.  Pure host code.
bool prf_vertical_plan(u32 kmin, u32 kmax, u32 min_repeats, u32 min_span, prf_vplan *plan) {
    if (kmin < 1 || kmax < kmin || kmax > PRF_VMAX_K || min_repeats < 2) return false;
    struct Item {
        prf_vtask t;
        u32 cost;
    };
    std::vector<Item> items;
    u32 reach = 0;       // furthest row of a lane's extended stream any task reads
    u32 covered_to = 0;  // group chunks cover motif sizes below this
    for (u32 k = kmin; k <= kmax; k++) {
        const long long M = prf_plan_min_matches(k, min_repeats, min_span);
        if (M < SMALL_M) {
            Item it;
            it.t.k0 = (unsigned short)k;   // <= 14: M >= (min_repeats - 1) * k >= k
            it.t.kind = (unsigned char)M;  // M >= 1 because min_repeats >= 2
            it.t.valid = 1;
            it.t.stride = 1;
            // measured (stamps build, 6 workgroups per CU, units of 10 cycles): 4.9 k cycles for M <= 6 (one operation per
            // window), 5.4 k for M = 7 .. 9 (two), 6.0 k from M = 10 on (more rows of the next lane)
            // measured on the final kernel (stamps build, 6 workgroups per CU, units of 10 cycles): 6.6 k cycles for the exact form
            // (M <= 6), 4.6 - 5.2 k for groups of 2 rows (M 7 - 10), 4.0 k for groups of 4 (M >= 11)
            it.cost = M <= 8 ? 660u : (M <= 10 ? 520u : 400u);
            items.push_back(it);
            reach = std::max<u32>(reach, 4 * (((u32)T + (u32)M - 1 + k + 3) / 4) - 1);
        } else if (k >= covered_to) {
            // A group task takes the motif sizes k0 .. k0+7 (k0 a multiple of 4: whole 16-byte slots), or only the first four of
            // them where the two halves want different strides (default thresholds: 8-11 every group, 12-19 every 2nd, 20..
            // every 4th).  Examine every group, every 2nd or every 4th: a run of >= 8*S + 7 positions contains an aligned group
            // of 8 whose index is a multiple of S.
            const u32 k0 = k & ~3u;
            u32 valid = 0;
            long long mmin[2] = {1ll << 40, 1ll << 40};
            for (u32 kk = 0; kk < 8; kk++) {
                const u32 kx = k0 + kk;
                const long long Mx = prf_plan_min_matches(kx, min_repeats, min_span);
                if (kx >= kmin && kx <= kmax && Mx >= SMALL_M) {
                    valid |= 1u << kk;
                    mmin[kk >> 2] = std::min(mmin[kk >> 2], Mx);
                }
            }
            auto stride_of = [](long long m) { return m >= 39 ? 4u : (m >= 23 ? 2u : 1u); };
            const bool both = (valid & 0x0Fu) && (valid & 0xF0u);
            if (both && stride_of(mmin[0]) != stride_of(mmin[1])) valid &= 0x0Fu;  // the second half starts a task of its own
            const bool half = (valid & 0xF0u) == 0;
            const u32 stride = stride_of(half ? mmin[0] : std::min(mmin[0], mmin[1]));
            Item it;
            it.t.k0 = (unsigned short)k0;
            it.t.kind = 0;
            it.t.valid = (unsigned char)valid;
            it.t.stride = (unsigned char)stride;
            // measured (units of 10 cycles): 10.5 k / 5.1 k / 4.2 k cycles with 8 sizes, 2.2 k for stride 4 with 3; a task of four
            // sizes reads three quarters of the rows of one of eight
            // measured: stride 1 with 4 sizes 6.5 k cycles, stride 2 with 8 sizes 8.3 k, stride 4 with 8 / 7 sizes 4.4 - 4.9 k / 4.2 k
            const u32 n_sizes = (u32)__builtin_popcount(valid);
            it.cost = stride == 1 ? 155u + 125u * n_sizes : (stride == 2 ? 105u + 90u * n_sizes : 140u + 40u * n_sizes);
            items.push_back(it);
            reach = std::max<u32>(reach, 24 + k0 + (half ? 11 : 15));
            covered_to = k0 + (half ? 4 : 8);
        }
    }
    if (items.size() > PRF_VMAX_TASKS) return false;
    // longest-processing-time-first assignment to at most 4 waves
    const u32 nw = std::max<u32>(1, std::min<u32>(PRF_VMAX_WAVES, (u32)items.size()));
    std::vector<std::vector<Item>> bins(nw);
    std::vector<u32> load(nw, 0);
    std::vector<Item> sorted = items;
    std::stable_sort(sorted.begin(), sorted.end(), [](const Item &a, const Item &b) { return a.cost > b.cost; });
    // The ticket for the workgroup's next launch slot is an atomic whose value the compiler waits for on the spot (~3 k
    // cycles with a thousand workgroups drawing): it is dealt like a task, to the wave with the least other work -- or to
    // a wave without tasks, if there is one.  (Tried and dropped: the waves of a workgroup pulling tasks from one list
    // through an LDS counter at run time -- every wave's scan got 2-3 k cycles longer; the stride-1 group task cut in two
    // halves of four sizes -- a half costs three quarters of the whole, its four blocks are LDS latency, not arithmetic.)
    constexpr u32 TICKET_COST = 30;  // (round 3: the atomic's value is no longer waited for on the spot)
    bool ticket_dealt = nw < (u32)PRF_VMAX_WAVES;
    plan->ticket_wave = nw < (u32)PRF_VMAX_WAVES ? nw : 0;
    for (const Item &it : sorted) {
        if (!ticket_dealt && it.cost <= TICKET_COST) {
            plan->ticket_wave = (u32)(std::min_element(load.begin(), load.end()) - load.begin());
            load[plan->ticket_wave] += TICKET_COST;
            ticket_dealt = true;
        }
        const u32 w = (u32)(std::min_element(load.begin(), load.end()) - load.begin());
        bins[w].push_back(it);
        load[w] += it.cost;
    }
    if (!ticket_dealt) {
        plan->ticket_wave = (u32)(std::min_element(load.begin(), load.end()) - load.begin());
        load[plan->ticket_wave] += TICKET_COST;
    }
    // local improvement of the greedy deal: while the busiest wave can hand a task to, or swap a task with, another wave so that
    // the larger of the two loads drops, do it (a dozen tasks: the default plan goes from 1980 to 1895 units of 10 cycles)
    for (int round = 0; round < 64; round++) {
        const u32 hi = (u32)(std::max_element(load.begin(), load.end()) - load.begin());
        bool moved = false;
        for (u32 o = 0; o < nw && !moved; o++) {
            if (o == hi) continue;
            for (size_t i = 0; i < bins[hi].size() && !moved; i++) {
                const u32 ci = bins[hi][i].cost;
                if (std::max(load[hi] - ci, load[o] + ci) < load[hi]) {  // move
                    bins[o].push_back(bins[hi][i]);
                    bins[hi].erase(bins[hi].begin() + (long)i);
                    load[hi] -= ci;
                    load[o] += ci;
                    moved = true;
                    break;
                }
                for (size_t j = 0; j < bins[o].size(); j++) {
                    const u32 cj = bins[o][j].cost;
                    if (cj < ci && std::max(load[hi] - ci + cj, load[o] + ci - cj) < load[hi]) {  // swap
                        std::swap(bins[hi][i], bins[o][j]);
                        load[hi] = load[hi] - ci + cj;
                        load[o] = load[o] + ci - cj;
                        moved = true;
                        break;
                    }
                }
            }
        }
        if (!moved) break;
    }
    {
        // default: the short, latency-bound phases (stage, the record waves of the verify phase, rows) at priority 2, the scan
        // (long, plenty of independent arithmetic) and the flag waves (they wait at the barrier anyway) at 0.  Measured on
        // the default workload (tools/prio_sweep.sh, gpurun_out/prio_sweep*.txt): 0.775 ms without priorities, 0.748-0.757
        // with any setting that raises records and rows; random sequence (few candidates) is indifferent.
        // PRF_PRIO (diagnostic) overrides.
        static const u32 prio_cfg = getenv("PRF_PRIO") ? (u32)strtoul(getenv("PRF_PRIO"), nullptr, 0) : 0xA02u;
        plan->prio = prio_cfg;
        const u32 busiest = *std::max_element(load.begin(), load.end());
        plan->slack_waves = 0;
        for (u32 w = 0; w < (u32)PRF_VMAX_WAVES; w++)
            if (w >= nw || 10u * load[w] < 9u * busiest) plan->slack_waves |= 1u << w;
    }
    plan->n_waves = nw;
    plan->n_tasks = 0;
    plan->n_group_k = 0;
    plan->n_exact = 0;
    // the motif sizes of the exact tasks are consecutive: M(k) = max((r-1) k, span - k) is V-shaped, so {k : M(k) < 15} is an interval
    u32 k_exact0 = ~0u, k_exact1 = 0, n_exact_items = 0;
    for (const Item &it : items)
        if (it.t.kind) {
            k_exact0 = std::min<u32>(k_exact0, it.t.k0);
            k_exact1 = std::max<u32>(k_exact1, it.t.k0);
            n_exact_items++;
        }
    if (n_exact_items && k_exact1 - k_exact0 + 1 != n_exact_items) return false;
    plan->k_exact0 = n_exact_items ? k_exact0 : 0;
    for (u32 w = 0; w < nw; w++) {
        plan->wave_begin[w] = plan->n_tasks;
        for (const Item &it : bins[w]) {
            prf_vtask t = it.t;
            t.pad = 0;
            t.item0 = 0;
            if (t.kind == 0) {
                t.item0 = (unsigned short)plan->n_group_k;
                plan->n_group_k += (u32)__builtin_popcount((unsigned)t.valid);
            } else {
                plan->n_exact++;
                t.item0 = (unsigned short)(t.k0 - k_exact0);
            }
            plan->tasks[plan->n_tasks++] = t;
        }
    }
    for (u32 w = nw; w <= PRF_VMAX_WAVES; w++) plan->wave_begin[w] = plan->n_tasks;
    const u32 need_nc = 64 + reach / T;  // row r of a lane's extended stream lies in virtual lane + r / 32
    plan->nc = need_nc <= 72 ? 72 : 80;  // the widths the kernel is instantiated for
    plan->cof_words = (kmax + 1 + 3) & ~3u;  // <= PRF_VMAX_K + 4: the table is declared with that many entries
    // header, R1 (image / window), records, row list, all-N masks, flag lists + counts, boundary items, cofactor table
    plan->lds_bytes = (u32)(SMEM_HDR + (size_t)2 * RG * plan->nc * 16 + (size_t)REC_CAP * sizeof(u64) +
                            (size_t)2 * ROW_CAP_LDS * sizeof(u32) + (size_t)SLOW_CAP * 16 + (size_t)FLAG_CAP * 2 +
                            (size_t)plan->n_group_k * sizeof(u32) + (size_t)plan->cof_words * sizeof(u32));
    plan->per_cu = 0;  // (set at the first launch: the occupancy the runtime reports for this much LDS)
    return need_nc <= 80;
}


int prf_plan_json(u32 kmin, u32 kmax, u32 min_repeats, u32 min_span, char *buf, u64 buf_len) {
    std::string out;
    prf_vplan plan;
    if (!prf_vertical_plan(kmin, kmax, min_repeats, min_span, &plan)) {
        out = "{\"path\": \"generic\"}";
    } else {
        char tmp[160];
        snprintf(tmp, sizeof tmp, "{\"path\": \"fused\", \"waves\": %u, \"nc\": %u, \"lds_bytes\": %u, \"tasks\": [", plan.n_waves,
                 plan.nc, plan.lds_bytes);
        out = tmp;
        bool first = true;
        for (u32 w = 0; w < plan.n_waves; w++)
            for (u32 ti = plan.wave_begin[w]; ti < plan.wave_begin[w + 1]; ti++) {
                const prf_vtask &t = plan.tasks[ti];
                snprintf(tmp, sizeof tmp, "%s{\"wave\": %u, \"kind\": %u, \"k0\": %u, \"valid\": %u, \"stride\": %u}", first ? "" : ", ",
                         w, (unsigned)t.kind, (unsigned)t.k0, (unsigned)t.valid, (unsigned)t.stride);
                out += tmp;
                first = false;
            }
        out += "]}";
    }
    if (out.size() + 1 > buf_len) return -1;
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}
