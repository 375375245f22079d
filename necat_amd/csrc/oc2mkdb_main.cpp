// oc2mkdb - drop-in for NECAT's volume writer (reference: makedb/main.c, common/packed_db.c:229-315).
//
//   oc2mkdb wrk-dir file-list [file-list ...]
//
// Every file list names FASTA / FASTQ files (plain or gzip), one per line.  Reads are packed 2 bits per base
// into volumes of >= 2 Gbp: `wrk-dir/vol<i>` (the PackedDB dump layout), `wrk-dir/volume_names.txt` (path, first
// read id, read count per volume) and `wrk-dir/reads_info.txt` (volumes, reads) - byte for byte what the
// reference writes, including its treatment of characters other than ACGT (nst_nt4_table codes 4 and 5 are OR-ed
// into the byte as they are, ontcns_aux.h:118).  This is host code: parsing is I/O bound; the device-side
// re-layout of a volume happens when it is uploaded (necat_volume_upload, k_repack).
//
// NECAT_MKDB_VOLSIZE overrides the 2 Gbp volume size (tests only; the reference's is a constant, makedb/main.c:8).
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <dlfcn.h>

#include <string>
#include <vector>

namespace {

const char kPacHeader[] = "ontcns_pac_header_hofuwhogfuewo";     // common/packed_db.c:7
constexpr int kPlatform = 0;                                      // TECH_PACBIO, makedb/main.c:23

// common/nst_nt4_table.c: A C G T (either case) -> 0..3, '-' -> 5, everything else -> 4
struct CodeTable {
    uint8_t t[256];
    CodeTable()
    {
        memset(t, 4, sizeof t);
        t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; t['-'] = 5;
    }
};
const CodeTable kCode;

// byte-stream reader with the buffer discipline of klib's kstream (16 KB blocks over gzread)
struct Stream {
    gzFile f = nullptr;
    std::vector<unsigned char> buf = std::vector<unsigned char>(16384);
    int begin = 0, end = 0;
    bool eof = false, err = false;
    int getc()
    {
        if (err) return -3;
        if (begin >= end) {
            if (eof) return -1;
            begin = 0;
            end = gzread(f, buf.data(), (unsigned)buf.size());
            if (end == 0) { eof = true; return -1; }
            if (end < 0) { eof = true; err = true; end = 0; return -3; }
        }
        return buf[begin++];
    }
    // ks_getuntil2 (klib/kseq.h:92-145): append bytes up to the delimiter (a line end, or any white space when
    // `space`), consume the delimiter, report it in *dret; < 0 when nothing could be read
    int until(bool space, std::string& s, int* dret, bool append)
    {
        bool gotany = false;
        if (dret) *dret = 0;
        if (!append) s.clear();
        for (;;) {
            if (err) return -3;
            if (begin >= end) {
                if (eof) break;
                begin = 0;
                end = gzread(f, buf.data(), (unsigned)buf.size());
                if (end == 0) { eof = true; break; }
                if (end < 0) { eof = true; err = true; end = 0; return -3; }
            }
            int i = begin;
            if (space) { while (i < end && !isspace(buf[i])) ++i; }
            else { while (i < end && buf[i] != '\n') ++i; }
            gotany = true;
            s.append((const char*)buf.data() + begin, (size_t)(i - begin));
            begin = i + 1;
            if (i < end) { if (dret) *dret = buf[i]; break; }
        }
        if (!gotany && eof && begin >= end) return -1;
        if (!space && s.size() > 1 && s.back() == '\r') s.pop_back();
        return (int)s.size();
    }
};

// kseq_read (klib/kseq.h:178-218): >= 0 sequence length, -1 end of file, -2 truncated quality string
struct Reader {
    Stream ks;
    int last_char = 0;
    std::string name, comment, seq, qual;
    int next()
    {
        int c;
        if (last_char == 0) {
            while ((c = ks.getc()) >= 0 && c != '>' && c != '@') {}
            if (c < 0) return c;
            last_char = c;
        }
        comment.clear(); seq.clear(); qual.clear();
        int r = ks.until(true, name, &c, false);
        if (r < 0) return r;
        if (c != '\n') ks.until(false, comment, nullptr, false);
        while ((c = ks.getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            seq.push_back((char)c);
            ks.until(false, seq, nullptr, true);
        }
        if (c == '>' || c == '@') last_char = c;
        if (c != '+') return (int)seq.size();
        while ((c = ks.getc()) >= 0 && c != '\n') {}
        if (c == -1) return -2;
        while (ks.until(false, qual, nullptr, true) >= 0 && qual.size() < seq.size()) {}
        last_char = 0;
        if (seq.size() != qual.size()) return -2;
        return (int)seq.size();
    }
};

struct SeqInfo { uint64_t offset, size, hdr_offset; int32_t platform; int32_t pad; };   // SequenceInfo, common/packed_db.h:12-18
static_assert(sizeof(SeqInfo) == 32, "SequenceInfo layout");

// NECAT_MKDB_GPU=1: the 2-bit packing runs on the GPU (necat_volume_pack of libnecat_hip.so, opened at run time so that this
// program stays usable on a host without one): the volume's text is kept and packed in one go when the volume is written.
struct GpuPacker {
    void* lib = nullptr; void* ctx = nullptr;
    int (*create)(int, void**) = nullptr; void (*destroy)(void*) = nullptr; const char* (*last_error)(const void*) = nullptr;
    int (*pack)(void*, const char*, uint64_t, const uint64_t*, const uint64_t*, uint64_t, uint8_t*, void**) = nullptr;
    bool open(std::string* err)
    {
        char self[4096]; std::string dir = ".";
        const ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
        if (n > 0) { self[n] = 0; dir = self; dir = dir.substr(0, dir.find_last_of('/')); }
        lib = dlopen((dir + "/libnecat_hip.so").c_str(), RTLD_NOW);
        if (!lib) lib = dlopen("libnecat_hip.so", RTLD_NOW);
        if (!lib) { *err = std::string("cannot open libnecat_hip.so: ") + dlerror(); return false; }
        *(void**)&create = dlsym(lib, "necat_ctx_create"); *(void**)&destroy = dlsym(lib, "necat_ctx_destroy");
        *(void**)&last_error = dlsym(lib, "necat_last_error"); *(void**)&pack = dlsym(lib, "necat_volume_pack");
        if (!create || !destroy || !last_error || !pack) { *err = "libnecat_hip.so lacks necat_volume_pack"; return false; }
        const char* dev = getenv("NECAT_GPU");
        if (create(dev ? atoi(dev) : 0, &ctx)) { *err = "no usable gfx950 device"; return false; }
        return true;
    }
    ~GpuPacker() { if (ctx && destroy) destroy(ctx); }
};

struct Volume {
    std::vector<uint8_t> pac;
    std::string text;                 // NECAT_MKDB_GPU=1: the bases as read, packed by the device at dump time
    GpuPacker* gpu = nullptr;
    uint64_t nbases = 0;
    std::vector<SeqInfo> info;
    std::string hdr;
    void add(const std::string& name, const std::string& seq)         // pdb_add_one_seq, packed_db.c:229-252
    {
        SeqInfo si; memset(&si, 0, sizeof si);
        si.offset = nbases; si.size = seq.size(); si.hdr_offset = hdr.size(); si.platform = kPlatform;
        info.push_back(si);
        hdr.append(name); hdr.push_back('\0');
        if (gpu) { text.append(seq); nbases += seq.size(); return; }
        if (pac.size() < (nbases + seq.size() + 3) / 4 + 1) pac.resize(((nbases + seq.size() + 3) / 4 + 1) * 2, 0);
        // _set_pac: the code is OR-ed in at the base's 2-bit slot, first base in the top bits; codes 4 and 5 spill
        // into the neighbouring slot (or out of the byte) exactly as in the reference
        const unsigned char* p = (const unsigned char*)seq.data();
        size_t i = 0, n = seq.size();
        uint8_t* out = pac.data();
        for (; i < n && (nbases & 3); ++i, ++nbases) out[nbases >> 2] = (uint8_t)(out[nbases >> 2] | (kCode.t[p[i]] << ((~nbases & 3) << 1)));
        for (; i + 4 <= n; i += 4, nbases += 4)              // whole bytes: same ORs, one store
            out[nbases >> 2] = (uint8_t)((kCode.t[p[i]] << 6) | (kCode.t[p[i + 1]] << 4) | (kCode.t[p[i + 2]] << 2) | kCode.t[p[i + 3]]);
        for (; i < n; ++i, ++nbases) out[nbases >> 2] = (uint8_t)(out[nbases >> 2] | (kCode.t[p[i]] << ((~nbases & 3) << 1)));
    }
    bool dump(const std::string& path)                                  // pdb_dump, packed_db.c:291-315
    {
        if (gpu) {
            pac.assign((nbases + 3) / 4 + 8, 0);
            if (gpu->pack(gpu->ctx, text.data(), nbases, nullptr, nullptr, 0, pac.data(), nullptr)) {
                fprintf(stderr, "necat_volume_pack: %s\n", gpu->last_error(gpu->ctx));
                return false;
            }
        }
        FILE* out = fopen(path.c_str(), "wb");
        if (!out) return false;
        const uint64_t ns = info.size(), hs = hdr.size(), pb = (nbases + 3) >> 2;
        bool ok = fwrite(kPacHeader, 1, strlen(kPacHeader), out) == strlen(kPacHeader) && fwrite(&ns, 8, 1, out) == 1 &&
                  fwrite(&nbases, 8, 1, out) == 1 && (ns == 0 || fwrite(info.data(), sizeof(SeqInfo), ns, out) == ns) &&
                  fwrite(&hs, 8, 1, out) == 1 && (hs == 0 || fwrite(hdr.data(), 1, hs, out) == hs) &&
                  (pb == 0 || fwrite(pac.data(), 1, pb, out) == pb);
        ok = fclose(out) == 0 && ok;
        return ok;
    }
    void clear() { std::fill(pac.begin(), pac.end(), 0); nbases = 0; info.clear(); hdr.clear(); text.clear(); }
};

std::string in_dir(const char* wrk_dir, const char* leaf)              // copy_wrk_dir_name, makedb_aux.c:6-10
{
    std::string s(wrk_dir);
    if (s.empty() || s.back() != '/') s.push_back('/');
    return s + leaf;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 3) {
        fprintf(stderr, "USAGE:\n%s wrk-dir file-list [file-list]\n", argv[0]);
        return 1;
    }
    const char* wrk_dir = argv[1];
    uint64_t vol_size = 2000000000ULL;                                  // kVolSize, makedb/main.c:8
    if (const char* e = getenv("NECAT_MKDB_VOLSIZE")) vol_size = strtoull(e, nullptr, 10);
    if (access(wrk_dir, F_OK) == -1 && mkdir(wrk_dir, 0755) == -1) { fprintf(stderr, "failed to create folder %s\n", wrk_dir); return 1; }
    FILE* vn_out = fopen(in_dir(wrk_dir, "volume_names.txt").c_str(), "w");
    if (!vn_out) { fprintf(stderr, "cannot write %s\n", in_dir(wrk_dir, "volume_names.txt").c_str()); return 1; }
    Volume vol;
    GpuPacker packer;
    if (const char* e = getenv("NECAT_MKDB_GPU")) if (atoi(e)) {
        std::string err;
        if (!packer.open(&err)) { fprintf(stderr, "NECAT_MKDB_GPU: %s\n", err.c_str()); return 1; }      // asked for the GPU: no silent host packing
        vol.gpu = &packer;
    }
    int vid = 0, num_reads = 0, read_start_id = 0;
    auto flush = [&]() -> bool {
        const std::string vname = in_dir(wrk_dir, ("vol" + std::to_string(vid)).c_str());
        if (!vol.dump(vname)) { fprintf(stderr, "cannot write %s\n", vname.c_str()); return false; }
        fprintf(vn_out, "%s\t%d\t%lu\n", vname.c_str(), read_start_id, (unsigned long)vol.info.size());
        read_start_id += (int)vol.info.size();
        vol.clear();
        ++vid;
        return true;
    };
    for (int a = 2; a < argc; ++a) {
        fprintf(stdout, "file_list: %s\n", argv[a]);
        FILE* lst = fopen(argv[a], "r");
        if (!lst) { fprintf(stderr, "cannot open %s\n", argv[a]); return 1; }
        char line[2048];
        while (fgets(line, sizeof line, lst)) {                         // pack_one_list, makedb/main.c:49-65
            size_t n = strlen(line);
            if (n && line[n - 1] == '\n') line[n - 1] = '\0';
            Reader rd;
            rd.ks.f = gzopen(line, "r");
            if (!rd.ks.f) { fprintf(stderr, "cannot open %s\n", line); return 1; }
            int n_file = 0; uint64_t bp_file = 0;
            uint64_t cvs = vol.nbases;                                   // pack_one_file, makedb/main.c:11-46
            while (rd.next() >= 0) {
                vol.add(rd.name, rd.seq);
                ++n_file; bp_file += rd.seq.size(); cvs += rd.seq.size();
                if (cvs >= vol_size) { if (!flush()) return 1; cvs = 0; }
            }
            gzclose(rd.ks.f);
            fprintf(stderr, "pack %s: %d reads, %lu bps\n", line, n_file, (unsigned long)bp_file);
            num_reads += n_file;
        }
        fclose(lst);
    }
    if (vol.nbases && !flush()) return 1;
    fclose(vn_out);
    FILE* ri = fopen(in_dir(wrk_dir, "reads_info.txt").c_str(), "w");    // dump_reads_info, makedb_aux.c:34-42
    if (!ri) { fprintf(stderr, "cannot write reads_info.txt\n"); return 1; }
    fprintf(ri, "%d\t%d\n", vid, num_reads);
    fclose(ri);
    return 0;
}
