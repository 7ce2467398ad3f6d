#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/turbo_front.h"
int main(int argc, char** argv) {
  for (int i = 1; i < argc; ++i) {
    char err[512] = {0};
    std::string p = argv[i];
    tf_model* m = p.size() > 4 && p.substr(p.size() - 4) == ".xml" ? tf_load_xcsp3(argv[i], err, sizeof(err)) : tf_load_fzn(argv[i], err, sizeof(err));
    if (!m) { std::printf("%s: %s\n", argv[i], err); continue; }
    int32_t st[9];
    tf_push_eps_strategy(m, 0, 0);
    tf_shuffle_strategy(m, 0, 3);
    for (int r = 0; r < 3; ++r) tf_simplify(m, nullptr, st);
    std::vector<tb_itv> s(tf_store(m), tf_store(m) + tf_num_vars(m));
    for (auto& d : s) d.ub = d.lb;
    int n = tf_format_solution(m, s.data(), nullptr, 0);
    std::string buf((size_t)n + 1, 0);
    tf_format_solution(m, s.data(), buf.data(), n + 1);
    std::vector<tb_itv> full((size_t)tf_original_num_vars(m));
    tf_expand_solution(m, s.data(), full.data());
    std::printf("%s: V %d->%d P %d->%d out %d bytes, %zu stat bytes\n", argv[i], st[0], tf_num_vars(m), st[1], tf_num_props(m), n, std::strlen(tf_fcn_statistics(m)));
    tf_free(m);
  }
  return 0;
}
