// oc2pcan - drop-in for NECAT's candidate partitioner (reference: partition_candidates/pcan.c,
// pcan_aux.c, pcan_options.c), the stage between `oc2pmov -j 0 -u 1` and `oc2cns`.
//
//   oc2pcan [-p batch_size] [-f files_open_at_once] [-t threads] wrk-dir candidates
//
// `candidates` = the concatenated 28-byte PackedGappedCandidate records of all oc2pmov jobs.  Reads are grouped in
// batches of batch_size consecutive ids; partition i receives every record whose TEMPLATE (subject) lies in batch i,
// after each record has also been offered with query and subject exchanged (change_pcan_roles,
// common/gapped_candidate.c:54-69) - a read is a template for all reads it overlaps, whichever side found the
// pair.  Output: candidates.p<i> for every batch (empty files included) and candidates.partitions (their number).
// The order of records inside a partition file is unspecified in the reference (worker threads append chunks); the
// record multisets are identical.  Host code: the stage is pure I/O (one sequential read per group of open files).
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace {

struct Rec { uint32_t w[7]; };       // PackedGappedCandidate, common/gapped_candidate.h:64-66
static_assert(sizeof(Rec) == 28, "record size");

// change_pcan_roles: the strand flags swap places, the anchor flag and the score stay, ids and ranges swap
inline Rec swapped(const Rec& s)
{
    Rec d;
    d.w[0] = (s.w[0] & ((1u << 30) - 1)) | ((s.w[0] >> 31) << 30) | (((s.w[0] >> 30) & 1u) << 31);
    d.w[1] = s.w[4]; d.w[2] = s.w[5]; d.w[3] = s.w[6];
    d.w[4] = s.w[1]; d.w[5] = s.w[2]; d.w[6] = s.w[3];
    return d;
}

std::string in_dir(const char* wrk_dir, const char* leaf)
{
    std::string s(wrk_dir);
    if (s.empty() || s.back() != '/') s.push_back('/');
    return s + leaf;
}

void usage(const char* prog)
{
    fprintf(stderr, "USAGE:\n%s [options] wrk_dir candidates\n\nOPTIONS AND DESCRIPTIONS:\n"
                    "-p <Integer> batch size\n-f <Integer> number of partition files\n-t <Integer> number threads\n\n"
                    "DEFAULT OPTIONS:\n-p 100000 -f 100 -t 1 \n", prog);
}

}  // namespace

int main(int argc, char** argv)
{
    int batch_size = 100000, files_at_once = 100, threads = 1;     // pcan_options.c:10-14
    if (argc < 3) { usage(argv[0]); return 1; }
    int c;
    opterr = 0;
    while ((c = getopt(argc - 2, argv, "p:f:t:")) != -1) {
        switch (c) {
            case 'p': batch_size = atoi(optarg); break;
            case 'f': files_at_once = atoi(optarg); break;
            case 't': threads = atoi(optarg); break;
            default: fprintf(stderr, "invalid option\n"); usage(argv[0]); return 1;
        }
    }
    (void)threads;
    if (batch_size < 1 || files_at_once < 1) { fprintf(stderr, "batch size and number of files must be positive\n"); return 1; }
    const char* wrk_dir = argv[argc - 2];
    const char* can_path = argv[argc - 1];
    int num_volumes = 0, num_reads = 0;
    {
        FILE* f = fopen(in_dir(wrk_dir, "reads_info.txt").c_str(), "r");       // load_num_reads, makedb_aux.c:58-68
        if (!f || fscanf(f, "%d%d", &num_volumes, &num_reads) != 2) { fprintf(stderr, "cannot read %s\n", in_dir(wrk_dir, "reads_info.txt").c_str()); return 1; }
        fclose(f);
    }
    const int num_batches = (num_reads + batch_size - 1) / batch_size;       // pcan.c:111
    {
        FILE* f = fopen((std::string(can_path) + ".partitions").c_str(), "w");   // dump_num_partitions, pcan_aux.c:42-51
        if (!f) { fprintf(stderr, "cannot write %s.partitions\n", can_path); return 1; }
        fprintf(f, "%d\n", num_batches);
        fclose(f);
    }
    const size_t kChunk = 1 << 20;                                            // records per read
    std::vector<Rec> in(kChunk);
    for (int sfid = 0; sfid < num_batches; sfid += files_at_once) {
        const int efid = sfid + files_at_once < num_batches ? sfid + files_at_once : num_batches;
        const int64_t lo = (int64_t)sfid * batch_size, hi = (int64_t)efid * batch_size;   // [min_read_id, max_read_id), pcan.c:125-126
        std::vector<FILE*> out(efid - sfid, nullptr);
        std::vector<std::vector<Rec>> buf(efid - sfid);
        for (int i = sfid; i < efid; ++i) {
            out[i - sfid] = fopen((std::string(can_path) + ".p" + std::to_string(i)).c_str(), "wb");
            if (!out[i - sfid]) { fprintf(stderr, "cannot write %s.p%d\n", can_path, i); return 1; }
        }
        FILE* cin = fopen(can_path, "rb");
        if (!cin) { fprintf(stderr, "cannot open %s\n", can_path); return 1; }
        auto in_range = [&](uint32_t id) { return (int64_t)(int32_t)id >= lo && (int64_t)(int32_t)id < hi; };
        bool wok = true;          // every write is checked: a full disk must not leave short candidates.p<i> files behind exit code 0
        auto put = [&](const Rec& r) {
            std::vector<Rec>& b = buf[(size_t)(((int64_t)(int32_t)r.w[1] - lo) / batch_size)];
            b.push_back(r);
            if (b.size() >= 65536) { wok = fwrite(b.data(), sizeof(Rec), b.size(), out[&b - buf.data()]) == b.size() && wok; b.clear(); }
        };
        size_t n;
        while ((n = fread(in.data(), sizeof(Rec), kChunk, cin)) > 0) {
            for (size_t i = 0; i < n; ++i) {                                   // pcan_func, pcan.c:47-75
                const Rec& r = in[i];
                const bool s_in = in_range(r.w[1]), q_in = in_range(r.w[4]);
                if (s_in) put(r);                 // the subject is a template of this group: the record as it is
                if (q_in) put(swapped(r));        // the query is one: the record with the roles exchanged
            }
        }
        const bool rok = !ferror(cin);
        fclose(cin);
        bool ok = wok && rok;
        for (size_t k = 0; k < buf.size(); ++k) {
            if (!buf[k].empty()) ok = fwrite(buf[k].data(), sizeof(Rec), buf[k].size(), out[k]) == buf[k].size() && ok;
            ok = fclose(out[k]) == 0 && ok;
        }
        if (!ok) { fprintf(stderr, "write error on %s.p*\n", can_path); return 1; }
    }
    return 0;
}
