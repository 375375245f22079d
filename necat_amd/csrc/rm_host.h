// rm_host.h - host side of necat_map_reference: rm_extend_candidates / rm_extend_candidate (reference_mapping/rm_worker.c:61-196)
// replayed on one read's candidates after the device aligned every one of them block-wise against its stretch of the reference
// (rm_window, ext_core.h).  In candidate order, as the reference walks them: a candidate whose anchor lies inside a record already
// accepted for this read is skipped (map_aux.c:4-20); one whose block-wise alignment failed is dropped; one whose alignment stops
// more than 500 bp short of its chained range goes through the rescue pair (rescue.h) on its stretch and is dropped if that fails.
// No HIP in this header.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/necat_hip.h"
#include "dev_common.h"
#include "ext_core.h"
#include "rescue.h"

namespace necat {
namespace rm {

// a volume's bases on the host: 2-bit words as on the device (base i in bits 2 (i & 31) of word i >> 5) + the sequence offsets
struct Words {
    const uint64_t* w = nullptr;
    const uint64_t* seq_off = nullptr;
    // bases [from, to) of sequence `id`, or of its reverse complement (coordinates on that strand)
    void decode(int64_t id, int rev, int64_t from, int64_t to, std::vector<uint8_t>& dst) const
    {
        const uint64_t b = seq_off[id], n = seq_off[id + 1] - b;
        dst.resize((size_t)(to - from));
        for (int64_t i = from; i < to; ++i) {
            const uint64_t g = rev ? b + n - 1 - (uint64_t)i : b + (uint64_t)i;
            const uint8_t c = (uint8_t)((w[g >> 5] >> ((g & 31) * 2)) & 3);
            dst[(size_t)(i - from)] = rev ? (uint8_t)(3 - c) : c;
        }
    }
};

// rm_worker.c:82, :103-110: is the block-wise alignment (record m, REV already turned to forward coordinates) short of the chain?
inline bool needs_rescue(const necat_candidate& c, const necat_m4& m)
{
    const bool check = c.qoff >= c.qbeg && c.qoff < c.qend && c.soff >= c.sbeg && c.soff < c.send;
    if (!check) return false;
    const int64_t qb = c.qdir ? (int64_t)(m.qsize - m.qend) : (int64_t)m.qoff, qe = c.qdir ? (int64_t)(m.qsize - m.qoff) : (int64_t)m.qend;
    const int64_t lhang = qb > (int64_t)c.qbeg ? qb - (int64_t)c.qbeg : 0, rhang = qe < (int64_t)c.qend ? (int64_t)c.qend - qe : 0;
    return lhang + rhang > 500;
}

struct Worker {
    rescue::Dalign dal;
    rescue::EdlibGo edl;
    std::vector<uint8_t> q, t;
    int32_t q_id = -1, q_dir = -1;
    uint64_t n_rescue_tried = 0, n_rescued = 0;
    Worker(const rescue::DalignSpec& spec, double error) : dal(spec), edl(error) {}

    // candidates [lo, hi) = one read's, in examination order; accepted records are appended to out
    void replay(const necat_candidate* c, const necat_m4* m4, const uint8_t* ok, uint64_t lo, uint64_t hi, const Words& reads, const Words& ref,
                int read_start_id, int ref_start_id, int min_align_size, std::vector<necat_m4>& out)
    {
        const size_t first = out.size();
        q_id = -1;
        for (uint64_t i = lo; i < hi; ++i) {
            const necat_candidate& cc = c[i];
            bool contained = false;
            for (size_t j = first; j < out.size() && !contained; ++j) {
                const necat_m4& m = out[j];
                contained = cc.qdir == m.qdir && cc.sid == m.sid && cc.qoff >= m.qoff && cc.qoff <= m.qend && cc.soff >= m.soff && cc.soff <= m.send;
            }
            if (contained || !ok[i]) continue;
            necat_m4 m = m4[i];
            if (needs_rescue(cc, m)) {
                ++n_rescue_tried;
                int64_t from, to, woff;
                rm_window((int64_t)cc.qoff, (int64_t)cc.qsize, (int64_t)cc.soff, (int64_t)cc.ssize, &from, &to, &woff);
                if (q_id != cc.qid || q_dir != cc.qdir) { reads.decode(cc.qid - read_start_id, cc.qdir, 0, (int64_t)cc.qsize, q); q_id = cc.qid; q_dir = cc.qdir; }
                ref.decode(cc.sid - ref_start_id, 0, from, to, t);
                if (!dal.go((const char*)q.data(), (int)cc.qoff, (int)cc.qsize, (const char*)t.data(), (int)woff, (int)(to - from), min_align_size)) continue;
                if (!edl.go((const char*)q.data(), dal.r.abpos, dal.r.aepos, (const char*)t.data(), dal.r.bbpos, dal.r.bepos, dal.r.diffs, min_align_size)) continue;
                ++n_rescued;
                m.qoff = (uint64_t)edl.qoff; m.qend = (uint64_t)edl.qend;
                m.soff = (uint64_t)(edl.toff + from); m.send = (uint64_t)(edl.tend + from);
                m.ident_perc = edl.ident_perc;
                if (cc.qdir == 1) { const uint64_t qo = m.qsize - m.qend, qe = m.qsize - m.qoff; m.qoff = qo; m.qend = qe; }
            }
            out.push_back(m);
        }
    }
};

}  // namespace rm
}  // namespace necat
