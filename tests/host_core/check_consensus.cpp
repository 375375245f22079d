// check_consensus.cpp - TEST ONLY.  Feeds the product's host-side consensus (necat_amd/csrc/cns_consensus.h) with the
// add_one_align calls the ORACLE's extension loop logs (full log: both gapped strings per call) and writes the two output
// files of oc2cns, so the consensus proper can be compared byte for byte with the reference's oc2cns on a machine without
// a GPU.  Nothing here is part of the product path.
//
// usage: check_consensus <wrk_dir> <partition file> <full log> <min_cov> <min_size> <full_consensus> <cns_out> <raw_out>
#include <algorithm>
#include <fstream>
#include <sstream>

#include "../../necat_amd/csrc/host_io.h"
#include "../../necat_amd/csrc/cns_consensus.h"

using namespace necat_host;

int main(int argc, char** argv)
{
    if (argc < 9) { fprintf(stderr, "usage\n"); return 2; }
    const char* wrk = argv[1];
    const int min_cov = atoi(argv[4]), min_size = atoi(argv[5]), full = atoi(argv[6]);
    std::string err;
    VolumesInfo vi;
    if (!load_volumes_info(wrk, &vi, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
    std::vector<std::vector<uint8_t>> reads; std::vector<std::string> names;
    for (int v = 0; v < vi.num_volumes; ++v) {
        HostVolume hv;
        if (!load_volume(vi.names[v].c_str(), &hv, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
        for (uint64_t i = 0; i < hv.offset.size(); ++i) {
            std::vector<uint8_t> r(hv.size[i]);
            for (uint64_t k = 0; k < hv.size[i]; ++k) { const uint64_t g = hv.offset[i] + k; r[k] = (uint8_t)((hv.pac[g >> 2] >> ((~g & 3) << 1)) & 3); }
            reads.push_back(std::move(r)); names.emplace_back(hv.name(i));
        }
    }
    // id range of the partition (sid = item[1] of the 28-byte records)
    int min_id = 1 << 30, max_id = -1;
    {
        std::ifstream in(argv[2], std::ios::binary);
        uint32_t item[7];
        while (in.read((char*)item, 28)) { min_id = std::min(min_id, (int)item[1]); max_id = std::max(max_id, (int)item[1]); }
    }
    std::string cns_txt, raw_txt;
    std::vector<uint8_t> corrected((size_t)std::max(0, max_id + 1), 0);
    struct Ov { std::vector<uint8_t> ops, q; int ncols, toff; double w; };
    std::vector<Ov> ovs;
    cns::Worker w;
    std::ifstream log(argv[3]);
    std::string line;
    while (std::getline(log, line)) {
        std::vector<std::string> f;
        { std::stringstream ss(line); std::string x; while (std::getline(ss, x, '\t')) f.push_back(x); }
        if (f.empty()) continue;
        if (f[0] == "A") {
            if (f.size() < 9) { fprintf(stderr, "the log must be a FULL log (gapped strings)\n"); return 2; }
            Ov o; o.toff = atoi(f[1].c_str()); o.w = strtod(f[3].c_str(), nullptr); o.ncols = atoi(f[4].c_str());
            const std::string &qa = f[7], &ta = f[8];
            o.ops.assign((size_t)(o.ncols + 3) / 4 + 1, 0);
            for (int i = 0; i < o.ncols; ++i) {
                const int op = qa[i] == '-' ? 2 : (ta[i] == '-' ? 1 : (qa[i] == ta[i] ? 0 : 3));
                o.ops[i >> 2] |= (uint8_t)(op << ((i & 3) * 2));
                if (qa[i] != '-') o.q.push_back((uint8_t)(qa[i] == 'A' ? 0 : qa[i] == 'C' ? 1 : qa[i] == 'G' ? 2 : 3));
            }
            ovs.push_back(std::move(o));
        } else if (f[0] == "T") {
            const int tid = atoi(f[1].c_str()), tsize = atoi(f[2].c_str());
            const double cutoff = strtod(f[3].c_str(), nullptr);
            const int num_can = atoi(f[4].c_str()), num_ovlps = atoi(f[5].c_str());
            std::vector<cns::OverlapIn> in;
            for (const Ov& o : ovs) {
                cns::OverlapIn x; x.ops = o.ops.data(); x.ncols = o.ncols; x.toff = o.toff; x.weight = o.w; x.qfwd = o.q.data(); x.qsize = (int)o.q.size(); x.qoff = 0; x.qdir = 0;
                in.push_back(x);
            }
            if ((int)reads[tid].size() != tsize) { fprintf(stderr, "template %d: size mismatch\n", tid); return 2; }
            const bool c = cns::consensus_template(w, in.data(), in.size(), reads[tid].data(), tsize, tid, names[tid].c_str(), min_cov, min_size, full != 0, num_can, num_ovlps,
                                                   cutoff, cns_txt, raw_txt);
            if (c) corrected[(size_t)tid] = 1;
            ovs.clear();
        }
    }
    for (int id = min_id; id < max_id; ++id)
        if (!corrected[(size_t)id]) cns::uncorrected_record(raw_txt, reads[id].data(), (int)reads[id].size(), id, names[id].c_str());
    std::ofstream(argv[7], std::ios::binary) << cns_txt;
    std::ofstream(argv[8], std::ios::binary) << raw_txt;
    return 0;
}
