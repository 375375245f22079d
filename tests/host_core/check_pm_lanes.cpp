// check_pm_lanes.cpp - the SOURCE of necat_amd/csrc/pm_job.h (pm_run_volume: a job's units on NECAT_PAIR_LANES lanes, records written in unit order) compiled with g++
// against a FAKE library: the C-ABI entry points the job calls are defined here - contexts are counters, a "mapping" sleeps (longest for the FIRST units, so that later
// units finish first), logs how many calls ran at the same time and on which context, and returns a few records whose ids name the (query volume, reference volume)
// pair.  CPU only; the real volume files of a project directory are read by the job as always.
//   check_pm_lanes <wrk-dir> <out-prefix>
// Checks, at 1 / 2 / 3 / 5 lanes and for both jobs (-j 1 M4 text, -j 0 candidate text):
//   * the job's file is the SAME BYTES in every lane mode, its records in unit order (query volumes ascending);
//   * with L lanes, min(L, units) mappings really overlapped, each lane on a context of its own (lane 0: the job's), lane l's units are l, l + L, ...;
//   * every volume / index the job made is freed, every result block handed out is released (necat_free), also after a failure;
//   * a unit that fails (the mapping of one query volume returns an error) makes the job return 1, print the error, leave no output file and no .part - and return
//     (no lane left waiting), whichever lane mode;
//   * a lane whose context cannot be created fails the job the same way.
#include <atomic>
#include <chrono>
#include <map>
#include <set>
#include "../../necat_amd/csrc/pm_job.h"

struct necat_ctx { int id; char err[128]; };
struct necat_volume { uint64_t nbases, nseq; int tag; };
struct necat_index { int k; };

namespace fake {
std::mutex mu;
std::atomic<int> live_ctx{0}, live_vol{0}, live_ix{0}, live_blocks{0}, running{0}, max_running{0}, created{0};
std::vector<std::pair<int, int>> calls;          // (context id, first read id of the query volume) of every mapping, in completion order
int fail_read_start = -1;                        // the mapping of the query volume that starts at this read id fails
int fail_ctx_create_after = -1;                  // necat_ctx_create fails once this many contexts exist
int delay_ms_first = 120;
void reset() { calls.clear(); max_running = 0; created = 0; }
void* block(size_t bytes) { ++live_blocks; return malloc(bytes ? bytes : 1); }
}  // namespace fake

extern "C" {
int necat_ctx_create(int, necat_ctx** out)
{
    if (fake::fail_ctx_create_after >= 0 && fake::live_ctx >= fake::fail_ctx_create_after) { *out = nullptr; return NECAT_ERR_DEVICE; }
    necat_ctx* c = new necat_ctx(); c->id = 100 + fake::created++; c->err[0] = 0; ++fake::live_ctx; *out = c; return NECAT_OK;
}
void necat_ctx_destroy(necat_ctx* c) { if (c) { --fake::live_ctx; delete c; } }
const char* necat_last_error(const necat_ctx* c) { return c ? c->err : "fake: no context for this lane"; }
int necat_volume_upload(necat_ctx*, const uint8_t*, uint64_t nbases, const uint64_t*, const uint64_t*, uint64_t nseq, necat_volume** out)
{
    necat_volume* v = new necat_volume(); v->nbases = nbases; v->nseq = nseq; v->tag = 0; ++fake::live_vol; *out = v; return NECAT_OK;
}
void necat_volume_free(necat_ctx*, necat_volume* v) { if (v) { --fake::live_vol; delete v; } }
int necat_index_build(necat_ctx*, const necat_volume*, int k, int, necat_index** out) { necat_index* ix = new necat_index(); ix->k = k; ++fake::live_ix; *out = ix; return NECAT_OK; }
void necat_index_free(necat_ctx*, necat_index* ix) { if (ix) { --fake::live_ix; delete ix; } }
void necat_free(void* p) { if (p) { --fake::live_blocks; free(p); } }

static int fake_map(necat_ctx* c, const necat_volume* reads, int read_start, int ref_start, int slot_lo, int slot_hi, int job, void** out, uint64_t* n_out)
{
    const int now = ++fake::running;
    for (int m = fake::max_running; now > m && !fake::max_running.compare_exchange_weak(m, now);) {}
    // the earlier the query volume, the longer its mapping: with several lanes the LATER units are ready first
    std::this_thread::sleep_for(std::chrono::milliseconds(std::max(10, fake::delay_ms_first - 25 * (read_start / 16))));
    --fake::running;
    { std::lock_guard<std::mutex> lk(fake::mu); fake::calls.emplace_back(c->id, read_start); }
    if (read_start == fake::fail_read_start) { snprintf(c->err, sizeof c->err, "fake: the pair of the query volume at read %d fails", read_start); return NECAT_ERR_DEVICE; }
    const uint64_t n = 3 + (uint64_t)(read_start % 4) + (uint64_t)(slot_hi - slot_lo == necat_host::kPmSlots ? 0 : 1);
    if (job == 1) {
        necat_m4* m = (necat_m4*)fake::block(n * sizeof(necat_m4));
        memset(m, 0, n * sizeof(necat_m4));
        for (uint64_t i = 0; i < n; ++i) {
            m[i].qid = read_start + (int)(i % reads->nseq); m[i].sid = ref_start; m[i].ident_perc = 80.0 + (double)i; m[i].vscore = (int)(10 + i);
            m[i].qoff = 1 + i; m[i].qend = 1000 + i; m[i].qsize = 5000; m[i].soff = 2; m[i].send = 999; m[i].ssize = 6000;
        }
        *out = m;
    } else {
        necat_candidate* cd = (necat_candidate*)fake::block(n * sizeof(necat_candidate));
        memset(cd, 0, n * sizeof(necat_candidate));
        for (uint64_t i = 0; i < n; ++i) { cd[i].qid = read_start + (int)(i % reads->nseq); cd[i].sid = ref_start; cd[i].score = (int)(20 + i); cd[i].qsize = 5000; cd[i].ssize = 6000; cd[i].qoff = 7 + i; cd[i].soff = 9; }
        *out = cd;
    }
    *n_out = n;
    return NECAT_OK;
}
int necat_map_pair(necat_ctx* c, const necat_index*, const necat_volume*, const necat_volume* reads, int rs, int fs, int, const necat_map_options*, int, necat_m4** out, uint64_t* n, uint64_t* nc)
{ if (nc) *nc = 0; return fake_map(c, reads, rs, fs, 0, necat_host::kPmSlots, 1, (void**)out, n); }
int necat_map_pair_part(necat_ctx* c, const necat_index*, const necat_volume*, const necat_volume* reads, int rs, int fs, int, const necat_map_options*, int, int, int lo, int hi, int,
                        necat_m4** out, uint64_t* n, uint64_t* nc)
{ if (nc) *nc = 0; return fake_map(c, reads, rs, fs, lo, hi, 1, (void**)out, n); }
int necat_find_candidates(necat_ctx* c, const necat_index*, const necat_volume*, const necat_volume* reads, int rs, int fs, int, const necat_map_options*, necat_candidate** out, uint64_t* n)
{ return fake_map(c, reads, rs, fs, 0, necat_host::kPmSlots, 0, (void**)out, n); }
int necat_find_candidates_part(necat_ctx* c, const necat_index*, const necat_volume*, const necat_volume* reads, int rs, int fs, int, const necat_map_options*, int, int lo, int hi, int,
                               necat_candidate** out, uint64_t* n)
{ return fake_map(c, reads, rs, fs, lo, hi, 0, (void**)out, n); }
int necat_pcan_partition(necat_ctx*, const necat_candidate*, uint64_t, int, int, uint32_t** recs, uint64_t** poff, int* np) { *recs = nullptr; *poff = nullptr; *np = 0; return NECAT_OK; }
void necat_default_options(necat_map_options* o) { memset(o, 0, sizeof *o); o->kmer_size = 15; o->num_threads = 1; }
}

using namespace necat_host;

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "check_pm_lanes: %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

static std::string slurp(const std::string& p) { FILE* f = fopen(p.c_str(), "rb"); if (!f) return std::string("<missing>"); std::string s; char b[4096]; size_t k; while ((k = fread(b, 1, sizeof b, f)) > 0) s.append(b, k); fclose(f); return s; }

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    VolumesInfo vi; std::string err;
    CHECK(load_volumes_info(argv[1], &vi, &err));
    CHECK(vi.num_volumes >= 4);                       // job 0 has at least four units
    const int V = vi.num_volumes;
    const PmTrace tr;
    for (int job = 0; job < 2; ++job) {
        necat_map_options opt; necat_default_options(&opt); opt.job = job; opt.use_hdr_as_id = 0; opt.binary_output = 0;
        std::string want;
        for (int L : {1, 2, 3, 5}) {
            setenv("NECAT_PAIR_LANES", std::to_string(L).c_str(), 1);
            fake::reset(); fake::fail_read_start = -1; fake::fail_ctx_create_after = -1;
            necat_ctx* ctx = nullptr; CHECK(necat_ctx_create(0, &ctx) == 0);
            const std::string out = std::string(argv[2]) + "_j" + std::to_string(job) + "_l" + std::to_string(L);
            int rc;
            {
                PmLanes lanes(0);
                CHECK(lanes.lanes == L && lanes.fixed);
                rc = pm_run_volume(ctx, vi, 0, opt, out.c_str(), "check", tr, nullptr, nullptr, &lanes);
                // lane l ran the units l, l + L, ...: every mapping of one context belongs to one residue class, lane 0 is the job's own context
                std::map<int, std::set<int>> units_of;
                for (auto& c : fake::calls) { int u = -1; for (int v = 0; v < V; ++v) if (vi.read_start_id[v] == c.second) u = v; CHECK(u >= 0); units_of[c.first].insert(u % std::min(L, V)); }
                CHECK((int)units_of.size() == std::min(L, V));
                for (auto& kv : units_of) CHECK(kv.second.size() == 1);
                CHECK(units_of.count(ctx->id) && *units_of[ctx->id].begin() == 0);
                CHECK(fake::live_ctx == std::min(L, V));              // the extra contexts live as long as the lanes object (the owner's next job reuses them)
            }
            CHECK(rc == 0);
            CHECK(fake::live_ctx == 1);
            CHECK((int)fake::calls.size() == V);
            CHECK(fake::max_running == std::min(L, V));               // the mappings really overlapped (and never more than the lanes)
            CHECK(fake::live_vol == 0 && fake::live_ix == 0 && fake::live_blocks == 0);
            const std::string got = slurp(out);
            CHECK(got != "<missing>" && !got.empty());
            CHECK(slurp(out + ".part") == "<missing>");
            if (want.empty()) want = got;
            CHECK(got == want);                                       // the same bytes whatever the lanes
            // records in unit order: the first column (qid) never falls back to an earlier query volume
            int last_vol = -1; size_t at = 0;
            while (at < got.size()) {
                const size_t nl = got.find('\n', at); CHECK(nl != std::string::npos);
                const int qid = atoi(got.c_str() + at);
                int v = -1; for (int q = 0; q < V; ++q) if (qid >= vi.read_start_id[q]) v = q;
                CHECK(v >= last_vol); last_vol = v; at = nl + 1;
            }
            CHECK(last_vol == V - 1);
            necat_ctx_destroy(ctx);
            CHECK(fake::live_ctx == 0);
        }
        // a failing unit (the third query volume), every lane mode: exit 1, nothing left behind, nobody left waiting
        for (int L : {1, 2, 3}) {
            setenv("NECAT_PAIR_LANES", std::to_string(L).c_str(), 1);
            fake::reset(); fake::fail_read_start = vi.read_start_id[2]; fake::fail_ctx_create_after = -1;
            necat_ctx* ctx = nullptr; CHECK(necat_ctx_create(0, &ctx) == 0);
            const std::string out = std::string(argv[2]) + "_fail_j" + std::to_string(job) + "_l" + std::to_string(L);
            int rc;
            { PmLanes lanes(0); rc = pm_run_volume(ctx, vi, 0, opt, out.c_str(), "check", tr, nullptr, nullptr, &lanes); }
            CHECK(rc == 1);
            CHECK(slurp(out) == "<missing>" && slurp(out + ".part") == "<missing>");
            CHECK(fake::live_vol == 0 && fake::live_ix == 0 && fake::live_blocks == 0);
            necat_ctx_destroy(ctx);
            CHECK(fake::live_ctx == 0);
        }
        // a lane without a context
        {
            setenv("NECAT_PAIR_LANES", "3", 1);
            fake::reset(); fake::fail_read_start = -1; fake::fail_ctx_create_after = 2;           // the job's context and lane 1's exist, lane 2's cannot be made
            necat_ctx* ctx = nullptr; CHECK(necat_ctx_create(0, &ctx) == 0);
            const std::string out = std::string(argv[2]) + "_noctx_j" + std::to_string(job);
            int rc;
            { PmLanes lanes(0); rc = pm_run_volume(ctx, vi, 0, opt, out.c_str(), "check", tr, nullptr, nullptr, &lanes); }
            CHECK(rc == 1);
            CHECK(slurp(out) == "<missing>" && slurp(out + ".part") == "<missing>");
            CHECK(fake::live_vol == 0 && fake::live_ix == 0 && fake::live_blocks == 0);
            necat_ctx_destroy(ctx);
            CHECK(fake::live_ctx == 0);
        }
    }
    // the default: two lanes for a job whose reference volume is small or that has enough units to pay for a second context's arenas, one otherwise
    unsetenv("NECAT_PAIR_LANES");
    { PmLanes lanes(0); CHECK(!lanes.fixed && lanes.for_job(1000, 2) == 2 && lanes.for_job(kPmLaneBases, 3) == 1 && lanes.for_job(2000000000ull, kPmLaneUnits - 1) == 1 && lanes.for_job(2000000000ull, kPmLaneUnits) == 2); }
    setenv("NECAT_PAIR_LANES", "1", 1);
    { PmLanes lanes(0); CHECK(lanes.fixed && lanes.for_job(1000, 100) == 1); }
    printf("ok\n");
    return 0;
}
