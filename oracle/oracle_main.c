/* oc2pmov-compatible command line around the oracle (TEST INFRASTRUCTURE ONLY).
 * argv contract: pm_one_volume/main.c:7-15,28-47. */
#include "necat_oracle.h"
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "USAGE:\n%s [options] wrk-dir volume-id output\n", argv[0]); return 1; }
    ora_options opt;
    ora_options_default(&opt);
    if (ora_options_parse(argc - 3, argv, &opt)) return 1;
    ora_stats st;
    int rc = ora_pm_main(&opt, atoi(argv[argc - 2]), argv[argc - 3], argv[argc - 1], &st);
    if (rc) return rc;
    fprintf(stdout, "oracle: records=%lu aligned_qbases=%lu index=%.3fs map=%.3fs\n",
            (unsigned long)st.n_records, (unsigned long)st.aligned_qbases, st.t_index, st.t_map);
    return 0;
}
