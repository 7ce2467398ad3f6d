#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/turbo_front.h"
extern "C" {
#include "../../oracle/oracle.h"
}
int main(int argc, char** argv) {
  for (int i = 1; i < argc; ++i) {
    char err[512] = {0};
    tf_model* m = tf_load_fzn(argv[i], err, sizeof(err));
    if (!m) { std::printf("%s: %s\n", argv[i], err); continue; }
    orc_config cfg; std::memset(&cfg, 0, sizeof(cfg));
    cfg.stop_after_n_solutions = 1; cfg.timeout_ms = 3000;
    std::vector<orc_itv> best((size_t)tf_num_vars(m));
    int32_t has = 0; orc_stats st;
    int rc = orc_solve(&cfg, tf_num_vars(m), (const orc_itv*)tf_store(m), tf_num_props(m), (const orc_prop*)tf_props(m), tf_num_strategies(m),
                       tf_strat_var_order(m), tf_strat_val_order(m), tf_strat_off(m), tf_strat_vars(m), tf_obj_var(m), best.data(), &has, &st);
    std::printf("%s rc=%d has=%d nodes=%llu\n", argv[i], rc, has, (unsigned long long)st.nodes);
    tf_free(m);
  }
}
