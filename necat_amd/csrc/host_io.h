// host_io.h - host-side file formats of the oc2pmov process boundary: option
// parsing, volume directory + 2-bit volume files in, candidate / M4 records out.
#pragma once
#include <ctype.h>
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>

#include <string>
#include <vector>

#include "../../include/necat_hip.h"

namespace necat_host {

// The process environment the command-line programs want from the HIP runtime, set before the first HIP call (the library itself never touches its host's
// environment): GPU_MAX_HW_QUEUES = 8 - the two lanes of the extension rounds are eight streams, and kernels of streams that share a hardware queue run one
// after the other (default 4: the second lane gains 3.7 % at yeast size, with 8 it gains 7.7 %).  A value the user has set is kept.
inline void necat_cli_env() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

// common/map_options.c:90-150 (flag string :10).  Flags that are not given keep the defaults of
// sDefaultPairwiseMapingOptions (the reference leaves them uninitialised in oc2pmov, main.c:34; the
// pipeline always passes all of them).
inline bool parse_options(int argc, char** argv, necat_map_options* o)
{
    optind = 1;
    int c;
    while ((c = getopt(argc, argv, "k:z:q:b:s:n:a:d:e:m:t:j:u:i:")) != -1) {
        switch (c) {
        case 'k': o->kmer_size = atoi(optarg); break;
        case 'z': o->scan_window = atoi(optarg); break;
        case 'q': o->kmer_cnt_cutoff = atoi(optarg); break;
        case 'b': o->block_size = atoi(optarg); break;
        case 's': o->block_score_cutoff = atoi(optarg); break;
        case 'n': o->num_candidates = atoi(optarg); break;
        case 'a': o->align_size_cutoff = atoi(optarg); break;
        case 'd': o->ddfs_cutoff = atof(optarg); break;
        case 'e': o->error = atof(optarg); break;
        case 'm': o->num_output = atoi(optarg); break;
        case 't': o->num_threads = atoi(optarg); break;
        case 'j': o->job = atoi(optarg); break;
        case 'u': o->binary_output = atoi(optarg); break;
        case 'i': o->use_hdr_as_id = atoi(optarg); break;
        default: return false;
        }
    }
    return true;
}

inline void describe_options(FILE* out, const necat_map_options* d)   // map_options.c:152-174
{
    fprintf(out, "-k <Integer>\tkmer size\n-z <Integer>\tscan window size\n-q <Integer>\tkmer occurs > q times will be ignored\n"
                 "-b <Integer>\tblock size\n-n <Integer>\tnumber of candidates\n-a <Integer>\tmin align length\n"
                 "-d <Real>\tddf score cutoff\n-e <Real>\tsequencing error\n-m <Integer>\tnumber of output\n"
                 "-t <Integer>\tnumber of cpu threads\n-j <0 or 1>\tjob: 0 = find candidates only, 1 = perform alignemnt\n"
                 "-u <0 or 1>\toutput binary results: 0 = no, 1 = yes\n-i <0 or 1>\tuse header as sequence id: 0 = no, 1 = yes\n\n"
                 "DEFAULT OPTIONS:\n");
    fprintf(out, "-k %d -z %d -q %d -b %d -s %d -n %d -a %d -d %f -e %f -m %d -t %d -j %d -u %d -i %d \n", d->kmer_size, d->scan_window,
            d->kmer_cnt_cutoff, d->block_size, d->block_score_cutoff, d->num_candidates, d->align_size_cutoff, d->ddfs_cutoff, d->error,
            d->num_output, d->num_threads, d->job, d->binary_output, d->use_hdr_as_id);
}

// MapOptions2String (map_options.c:70-87)
inline std::string options_to_string(const necat_map_options* p)
{
    char b[512];
    snprintf(b, sizeof b, "-k %d -z %d -q %d -b %d -s %d -n %d -a %d -d %f -e %f -m %d -t %d -j %d -u %d -i %d ", p->kmer_size, p->scan_window,
             p->kmer_cnt_cutoff, p->block_size, p->block_score_cutoff, p->num_candidates, p->align_size_cutoff, p->ddfs_cutoff, p->error,
             p->num_output, p->num_threads, p->job, p->binary_output, p->use_hdr_as_id);
    return b;
}

struct VolumesInfo {      // makedb_aux.h:23-31
    int num_volumes = 0, num_reads = 0;
    std::vector<std::string> names;
    std::vector<int> read_start_id, read_count;
};

inline std::string dir_prefix(const char* wrk_dir)
{
    std::string s(wrk_dir);
    if (s.empty() || s.back() != '/') s += '/';
    return s;
}

inline bool load_volumes_info(const char* wrk_dir, VolumesInfo* vi, std::string* err)   // makedb_aux.c:36-118
{
    const std::string base = dir_prefix(wrk_dir);
    FILE* in = fopen((base + "reads_info.txt").c_str(), "r");
    if (!in) { *err = "cannot open " + base + "reads_info.txt"; return false; }
    if (fscanf(in, "%d%d", &vi->num_volumes, &vi->num_reads) != 2) { fclose(in); *err = "bad reads_info.txt"; return false; }
    fclose(in);
    in = fopen((base + "volume_names.txt").c_str(), "r");
    if (!in) { *err = "cannot open " + base + "volume_names.txt"; return false; }
    char line[4096];
    for (int i = 0; i < vi->num_volumes; ++i) {
        if (!fgets(line, sizeof line, in)) { fclose(in); *err = "volume_names.txt is truncated"; return false; }
        size_t k = 0, n = strlen(line);
        while (k < n && !isspace((unsigned char)line[k])) ++k;
        vi->names.emplace_back(line, k);
        ++k;
        vi->read_start_id.push_back(atoi(line + k));
        while (k < n && !isspace((unsigned char)line[k])) ++k;
        ++k;
        vi->read_count.push_back(atoi(line + k));
    }
    fclose(in);
    return true;
}

struct HostVolume {       // PackedDB (packed_db.h:21-27) as read by pdb_load_pac (packed_db.c:317-345)
    std::vector<uint8_t> pac;
    std::vector<uint64_t> offset, size, hdr_offset;
    std::string hdr;
    uint64_t nbases = 0;
    const char* name(uint64_t i) const { return hdr.c_str() + hdr_offset[i]; }
};

inline bool load_volume(const char* path, HostVolume* v, std::string* err)
{
    static const char magic[] = "ontcns_pac_header_hofuwhogfuewo";   // packed_db.c:7
    FILE* in = fopen(path, "rb");
    if (!in) { *err = std::string("cannot open volume ") + path; return false; }
    char m[64];
    const size_t ml = strlen(magic);
    bool ok = fread(m, 1, ml, in) == ml && memcmp(m, magic, ml) == 0;
    uint64_t ns = 0, nb = 0, hb = 0;
    ok = ok && fread(&ns, 8, 1, in) == 1 && fread(&nb, 8, 1, in) == 1;
    // the header's sizes are checked against the file before anything is sized by them (a truncated or foreign file is an
    // error message and exit 1 like every other failure, not a bad_alloc)
    uint64_t fsize = 0;
    if (ok) { const long at = ftell(in); ok = at >= 0 && fseek(in, 0, SEEK_END) == 0; if (ok) { fsize = (uint64_t)ftell(in); ok = fseek(in, at, SEEK_SET) == 0; } }
    ok = ok && ns <= fsize / 32 && nb / 4 <= fsize;
    if (ok) {
        v->offset.resize(ns); v->size.resize(ns); v->hdr_offset.resize(ns);
        for (uint64_t i = 0; ok && i < ns; ++i) {
            uint64_t rec[4];
            ok = fread(rec, 32, 1, in) == 1;
            v->offset[i] = rec[0]; v->size[i] = rec[1]; v->hdr_offset[i] = rec[2];
        }
    }
    ok = ok && fread(&hb, 8, 1, in) == 1 && hb <= fsize;
    if (ok) { v->hdr.resize(hb); ok = hb == 0 || fread(&v->hdr[0], 1, hb, in) == hb; }
    // names are read through hdr_offset (-i 1): every one must start inside the header block, which must end in a NUL
    for (uint64_t i = 0; ok && i < ns; ++i) ok = v->hdr_offset[i] < hb;
    ok = ok && (ns == 0 || hb == 0 || v->hdr[hb - 1] == 0);
    if (ok) { v->pac.resize((nb + 3) / 4 + 8); ok = nb == 0 || fread(v->pac.data(), 1, (nb + 3) / 4, in) == (nb + 3) / 4; }
    fclose(in);
    if (!ok) { *err = std::string("Invalid pac format database: '") + path + "'"; return false; }
    v->nbases = nb;
    return true;
}

// number of bases of a volume file, from its header alone (packed_db.c:291-296: magic, nseq, nbases)
inline bool volume_bases(const char* path, uint64_t* nbases, std::string* err)
{
    FILE* in = fopen(path, "rb");
    if (!in) { *err = std::string("cannot open volume ") + path; return false; }
    unsigned char h[31 + 16];
    const bool ok = fread(h, 1, sizeof h, in) == sizeof h && memcmp(h, "ontcns_pac_header_hofuwhogfuewo", 31) == 0;
    fclose(in);
    if (!ok) { *err = std::string("Invalid pac format database: '") + path + "'"; return false; }
    memcpy(nbases, h + 31 + 8, 8);
    return true;
}

inline void pack_candidate(const necat_candidate* c, uint32_t item[7])   // gapped_candidate.c:13-30
{
    memset(item, 0, 28);
    if (c->sdir == 1) item[0] |= 1u << 31;
    if (c->qdir == 1) item[0] |= 1u << 30;
    if (c->qoff == c->qbeg) item[0] |= 1u << 29;
    item[0] |= (uint32_t)(c->score < 1000000 ? c->score : 1000000);
    item[1] = (uint32_t)c->sid; item[2] = (uint32_t)c->sbeg; item[3] = (uint32_t)c->send;
    item[4] = (uint32_t)c->qid; item[5] = (uint32_t)c->qbeg; item[6] = (uint32_t)c->qend;
}

inline void log_line(const char* fmt, const char* what, double secs = -1)   // OC_LOG / TIMING_* (ontcns_aux.h:107-116)
{
    time_t t = time(NULL);
    char tb[64];
    snprintf(tb, sizeof tb, "%s", ctime(&t));
    size_t n = strlen(tb);
    if (n && tb[n - 1] == '\n') tb[n - 1] = 0;
    if (secs < 0) fprintf(stdout, "[%s] INFO: '%s' BEGINS\n", tb, what);
    else fprintf(stdout, fmt, tb, what, secs);
    fflush(stdout);
}
inline double now_sec() { struct timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + 1e-6 * tv.tv_usec; }

}  // namespace necat_host
