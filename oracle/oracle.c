/*
 * oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).  See oracle.h.
 *
 * Sequential, deterministic, single-threaded (the reference CPU path is single threaded:
 * cpu_solving.hpp:12, src/config.cpp:22).
 *
 * What follows which reference lines:
 *   orc_deduce / orc_ask   : call sites cpu_solving.hpp:26,34; barebones_dive_and_solve.hpp:931,944,977;
 *                            operator conventions common_solving.hpp:739-771.  Bodies are in
 *                            lattice-land/lala-pc v1.2.8 (absent) -> restated from the published
 *                            bounds-propagation rules.  PARITY UNPINNED at this level.
 *   orc_propagate          : cpu_solving.hpp:26-35 (Gauss-Seidel fixpoint, then select(!ask)).
 *   split / push_decision  : barebones_dive_and_solve.hpp:187-405.
 *   solve loop             : cpu_solving.hpp:23-45 with the explicit decision stack of
 *                            barebones_dive_and_solve.hpp:656-886 (dive, skip, solve, ropes, recompute).
 *   bookkeeping            : common_solving.hpp:829-878, barebones_dive_and_solve.hpp:1016-1030.
 */
#define _POSIX_C_SOURCE 200809L
#include "oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

/* Optional sink for every accepted solution leaf (tests enumerate solutions with it); not thread safe. */
static orc_itv* g_sink = NULL;
static int64_t g_sink_cap = 0, g_sink_count = 0;
void orc_set_solution_sink(orc_itv* buf, int64_t capacity) { g_sink = buf; g_sink_cap = capacity; g_sink_count = 0; }
int64_t orc_solution_sink_count(void) { return g_sink_count; }

/* Optional per-node trace (tests pick a node budget whose last node did not fail) and copy of the store the search
 * stopped on (tests compare it with the engine's, tb_session_debug_last_store); not thread safe. */
static unsigned char* g_trace = NULL;
static int64_t g_trace_cap = 0;
static orc_itv* g_last_store = NULL;
void orc_set_node_trace(unsigned char* failed_flags, int64_t capacity) { g_trace = failed_flags; g_trace_cap = capacity; }
void orc_set_last_store_sink(orc_itv* buf) { g_last_store = buf; }
static orc_path_header* g_path_hdr = NULL;
static orc_path_decision* g_path_dec = NULL;
static int32_t g_path_cap = 0;
void orc_set_path_sink(orc_path_header* hdr, orc_path_decision* decisions, int32_t capacity) { g_path_hdr = hdr; g_path_dec = decisions; g_path_cap = capacity; }

#define NINF ORC_NINF
#define PINF ORC_PINF

/* ---------- extended 32-bit integer arithmetic (bounds with +-infinity sentinels) ---------- */

static inline int is_inf(int32_t a) { return a == NINF || a == PINF; }

static inline int32_t clamp64(int64_t v) {
  if (v >= (int64_t)PINF) return PINF;
  if (v <= (int64_t)NINF) return NINF;
  return (int32_t)v;
}

static inline int32_t neg_ext(int32_t a) { return a == NINF ? PINF : (a == PINF ? NINF : -a); }

/* Bounds of sums / differences.  An infinite bound on the side that matters absorbs, everything else
 * saturates (a bound on the "wrong" side -- lb = +inf, ub = -inf -- only occurs for empty domains). */
static inline int32_t add_lo(int32_t a, int32_t b) { /* lb(A+B) from lb(A), lb(B) */
  return (a == NINF || b == NINF) ? NINF : clamp64((int64_t)a + (int64_t)b);
}
static inline int32_t add_hi(int32_t a, int32_t b) { /* ub(A+B) from ub(A), ub(B) */
  return (a == PINF || b == PINF) ? PINF : clamp64((int64_t)a + (int64_t)b);
}
static inline int32_t sub_lo(int32_t a, int32_t b) { /* lb(A-B) from lb(A), ub(B) */
  return (a == NINF || b == PINF) ? NINF : clamp64((int64_t)a - (int64_t)b);
}
static inline int32_t sub_hi(int32_t a, int32_t b) { /* ub(A-B) from ub(A), lb(B) */
  return (a == PINF || b == NINF) ? PINF : clamp64((int64_t)a - (int64_t)b);
}
static inline int32_t sat_inc(int32_t a) { return clamp64((int64_t)a + 1); }
static inline int32_t sat_dec(int32_t a) { return clamp64((int64_t)a - 1); }

static inline int32_t mul_ext(int32_t a, int32_t b) {
  if (a == 0 || b == 0) return 0;
  if (is_inf(a) || is_inf(b)) return ((a < 0) != (b < 0)) ? NINF : PINF;
  return clamp64((int64_t)a * (int64_t)b);
}

static inline int64_t div_floor64(int64_t a, int64_t b) {
  int64_t q = a / b, r = a % b;
  return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}
static inline int64_t div_ceil64(int64_t a, int64_t b) {
  int64_t q = a / b, r = a % b;
  return (r != 0 && ((r < 0) == (b < 0))) ? q + 1 : q;
}

static inline int32_t min32(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t max32(int32_t a, int32_t b) { return a > b ? a : b; }
static inline int64_t min64(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }
static inline int64_t abs64(int64_t a) { return a < 0 ? -a : a; }

/* VStore::embed: intersect the domain of v with [l,u]; report change / emptiness. */
static inline void embed(orc_itv* store, int32_t v, int32_t l, int32_t u, int* changed, int* failed) {
  orc_itv* d = &store[v];
  if (l > d->lb) { d->lb = l; *changed = 1; }
  if (u < d->ub) { d->ub = u; *changed = 1; }
  if (d->lb > d->ub) *failed = 1;
}

int orc_deduce(const orc_prop* p, orc_itv* store, int* failed) {
  const orc_itv X = store[p->x], Y = store[p->y], Z = store[p->z];
  int changed = 0;
  if (X.lb > X.ub || Y.lb > Y.ub || Z.lb > Z.ub) { *failed = 1; return 0; }
  switch (p->op) {
    case ORC_ADD: {
      embed(store, p->x, add_lo(Y.lb, Z.lb), add_hi(Y.ub, Z.ub), &changed, failed);
      embed(store, p->y, sub_lo(X.lb, Z.ub), sub_hi(X.ub, Z.lb), &changed, failed);
      embed(store, p->z, sub_lo(X.lb, Y.ub), sub_hi(X.ub, Y.lb), &changed, failed);
      break;
    }
    case ORC_MUL: {
      int32_t c0 = mul_ext(Y.lb, Z.lb), c1 = mul_ext(Y.lb, Z.ub), c2 = mul_ext(Y.ub, Z.lb), c3 = mul_ext(Y.ub, Z.ub);
      embed(store, p->x, min32(min32(c0, c1), min32(c2, c3)), max32(max32(c0, c1), max32(c2, c3)), &changed, failed);
      int x_nz = (X.lb > 0 || X.ub < 0);
      if (x_nz) { /* a non-zero product has non-zero factors */
        if (Y.lb == 0) embed(store, p->y, 1, PINF, &changed, failed);
        if (Y.ub == 0) embed(store, p->y, NINF, -1, &changed, failed);
        if (Z.lb == 0) embed(store, p->z, 1, PINF, &changed, failed);
        if (Z.ub == 0) embed(store, p->z, NINF, -1, &changed, failed);
      }
      int x_fin = !is_inf(X.lb) && !is_inf(X.ub);
      if (x_fin && (Z.lb > 0 || Z.ub < 0) && !is_inf(Z.lb) && !is_inf(Z.ub)) {
        int64_t lo = min64(min64(div_ceil64(X.lb, Z.lb), div_ceil64(X.lb, Z.ub)), min64(div_ceil64(X.ub, Z.lb), div_ceil64(X.ub, Z.ub)));
        int64_t hi = max64(max64(div_floor64(X.lb, Z.lb), div_floor64(X.lb, Z.ub)), max64(div_floor64(X.ub, Z.lb), div_floor64(X.ub, Z.ub)));
        embed(store, p->y, clamp64(lo), clamp64(hi), &changed, failed);
      }
      if (x_fin && (Y.lb > 0 || Y.ub < 0) && !is_inf(Y.lb) && !is_inf(Y.ub)) {
        int64_t lo = min64(min64(div_ceil64(X.lb, Y.lb), div_ceil64(X.lb, Y.ub)), min64(div_ceil64(X.ub, Y.lb), div_ceil64(X.ub, Y.ub)));
        int64_t hi = max64(max64(div_floor64(X.lb, Y.lb), div_floor64(X.lb, Y.ub)), max64(div_floor64(X.ub, Y.lb), div_floor64(X.ub, Y.ub)));
        embed(store, p->z, clamp64(lo), clamp64(hi), &changed, failed);
      }
      break;
    }
    case ORC_TDIV:
    case ORC_TMOD: {
      /* the divisor is never 0 */
      int32_t zl = Z.lb, zu = Z.ub;
      if (zl == 0) { zl = 1; embed(store, p->z, 1, PINF, &changed, failed); }
      if (zu == 0) { zu = -1; embed(store, p->z, NINF, -1, &changed, failed); }
      if (zl > zu) break;
      int z_fin = !is_inf(zl) && !is_inf(zu);
      int y_fin = !is_inf(Y.lb) && !is_inf(Y.ub);
      int z_nz = (zl > 0 || zu < 0);
      if (p->op == ORC_TDIV) {
        if (z_nz && z_fin && y_fin) {
          int64_t q0 = (int64_t)Y.lb / zl, q1 = (int64_t)Y.lb / zu, q2 = (int64_t)Y.ub / zl, q3 = (int64_t)Y.ub / zu;
          embed(store, p->x, clamp64(min64(min64(q0, q1), min64(q2, q3))), clamp64(max64(max64(q0, q1), max64(q2, q3))), &changed, failed);
        } else if (y_fin) {
          int64_t m = max64(abs64(Y.lb), abs64(Y.ub));
          embed(store, p->x, clamp64(-m), clamp64(m), &changed, failed);
        }
        if (!is_inf(X.lb) && !is_inf(X.ub) && z_fin) {
          int64_t p0 = (int64_t)X.lb * zl, p1 = (int64_t)X.lb * zu, p2 = (int64_t)X.ub * zl, p3 = (int64_t)X.ub * zu;
          int64_t m = max64(abs64(zl), abs64(zu)) - 1;
          embed(store, p->y, clamp64(min64(min64(p0, p1), min64(p2, p3)) - m), clamp64(max64(max64(p0, p1), max64(p2, p3)) + m), &changed, failed);
        }
      } else {
        int32_t m = z_fin ? clamp64(max64(abs64(zl), abs64(zu)) - 1) : PINF;
        if (Y.lb >= 0) embed(store, p->x, 0, min32(m, Y.ub), &changed, failed);
        else if (Y.ub <= 0) embed(store, p->x, max32(neg_ext(m), Y.lb), 0, &changed, failed);
        else embed(store, p->x, neg_ext(m), m, &changed, failed);
        if (y_fin && Y.lb == Y.ub && z_fin && zl == zu) {
          int32_t r = (int32_t)((int64_t)Y.lb % (int64_t)zl);
          embed(store, p->x, r, r, &changed, failed);
        }
      }
      break;
    }
    case ORC_MIN: {
      embed(store, p->x, min32(Y.lb, Z.lb), min32(Y.ub, Z.ub), &changed, failed);
      embed(store, p->y, X.lb, PINF, &changed, failed);
      embed(store, p->z, X.lb, PINF, &changed, failed);
      if (Y.lb > X.ub) embed(store, p->z, NINF, X.ub, &changed, failed);
      if (Z.lb > X.ub) embed(store, p->y, NINF, X.ub, &changed, failed);
      break;
    }
    case ORC_MAX: {
      embed(store, p->x, max32(Y.lb, Z.lb), max32(Y.ub, Z.ub), &changed, failed);
      embed(store, p->y, NINF, X.ub, &changed, failed);
      embed(store, p->z, NINF, X.ub, &changed, failed);
      if (Y.ub < X.lb) embed(store, p->z, X.lb, PINF, &changed, failed);
      if (Z.ub < X.lb) embed(store, p->y, X.lb, PINF, &changed, failed);
      break;
    }
    case ORC_EQ: {
      if (X.lb >= 1) {
        embed(store, p->y, Z.lb, Z.ub, &changed, failed);
        embed(store, p->z, Y.lb, Y.ub, &changed, failed);
      } else if (X.ub <= 0) {
        if (Y.lb == Y.ub) {
          if (Z.lb == Y.lb) embed(store, p->z, sat_inc(Y.lb), PINF, &changed, failed);
          if (Z.ub == Y.lb) embed(store, p->z, NINF, sat_dec(Y.lb), &changed, failed);
        }
        if (Z.lb == Z.ub) {
          if (Y.lb == Z.lb) embed(store, p->y, sat_inc(Z.lb), PINF, &changed, failed);
          if (Y.ub == Z.lb) embed(store, p->y, NINF, sat_dec(Z.lb), &changed, failed);
        }
      } else {
        if (Y.ub < Z.lb || Y.lb > Z.ub) embed(store, p->x, NINF, 0, &changed, failed);
        else if (Y.lb == Y.ub && Z.lb == Z.ub && Y.lb == Z.lb) embed(store, p->x, 1, PINF, &changed, failed);
      }
      break;
    }
    case ORC_LEQ: {
      if (X.lb >= 1) {
        embed(store, p->y, NINF, Z.ub, &changed, failed);
        embed(store, p->z, Y.lb, PINF, &changed, failed);
      } else if (X.ub <= 0) {
        embed(store, p->y, add_lo(Z.lb, 1), PINF, &changed, failed);
        embed(store, p->z, NINF, add_hi(Y.ub, -1), &changed, failed);
      } else {
        if (Y.ub <= Z.lb) embed(store, p->x, 1, PINF, &changed, failed);
        else if (Y.lb > Z.ub) embed(store, p->x, NINF, 0, &changed, failed);
      }
      break;
    }
    default: *failed = 1; break;
  }
  return changed;
}

int orc_ask(const orc_prop* p, const orc_itv* store) {
  const orc_itv X = store[p->x], Y = store[p->y], Z = store[p->z];
  switch (p->op) {
    case ORC_EQ:
      return (X.lb >= 1 && Y.lb == Y.ub && Z.lb == Z.ub && Y.lb == Z.lb) || (X.ub <= 0 && (Y.ub < Z.lb || Y.lb > Z.ub));
    case ORC_LEQ:
      return (X.lb >= 1 && Y.ub <= Z.lb) || (X.ub <= 0 && Y.lb > Z.ub);
    default: break;
  }
  if (X.lb != X.ub || Y.lb != Y.ub || Z.lb != Z.ub) return 0;
  int64_t x = X.lb, y = Y.lb, z = Z.lb;
  if ((p->op == ORC_MUL || p->op == ORC_TDIV || p->op == ORC_TMOD) && (is_inf(X.lb) || is_inf(Y.lb) || is_inf(Z.lb))) return 0;
  switch (p->op) {
    case ORC_ADD: return x == y + z;
    case ORC_MUL: return x == y * z;
    case ORC_TDIV: return z != 0 && x == y / z;
    case ORC_TMOD: return z != 0 && x == y % z;
    case ORC_MIN: return x == (y < z ? y : z);
    case ORC_MAX: return x == (y > z ? y : z);
    default: return 0;
  }
}

static int fixpoint(int32_t n_props, const orc_prop* props, orc_itv* store, int failed_in, uint64_t* iterations, uint64_t* deductions) {
  int failed = failed_in;
  int changed = 1;
  while (changed && !failed) {
    changed = 0;
    for (int32_t i = 0; i < n_props; ++i) changed |= orc_deduce(&props[i], store, &failed);
    ++*iterations;
    *deductions += (uint64_t)n_props;
  }
  return failed;
}

int orc_propagate(int32_t n_vars, orc_itv* store, int32_t n_props, const orc_prop* props,
                  uint64_t* iterations, uint64_t* deductions, int* all_entailed) {
  int failed = 0;
  for (int32_t v = 0; v < n_vars; ++v)
    if (store[v].lb > store[v].ub) failed = 1;
  uint64_t it = 0, de = 0;
  failed = fixpoint(n_props, props, store, failed, &it, &de);
  if (iterations) *iterations = it;
  if (deductions) *deductions = de;
  int ent = 1;
  if (!failed)
    for (int32_t i = 0; i < n_props && ent; ++i) ent = orc_ask(&props[i], store);
  if (all_entailed) *all_entailed = failed ? 0 : ent;
  return failed;
}

/* ---------------------------------- search ---------------------------------- */

typedef struct { /* lala LightBranch, barebones:135,355-393 */
  int32_t var, cur;
  orc_itv child[2];
  int32_t rope[2];
  int32_t obj_ub; /* test aid (orc_set_path_sink): the objective's upper bound in force when the decision was taken */
} decision_t;

typedef struct {
  const orc_config* cfg;
  int32_t n_vars, n_props, n_strats, obj_var;
  const orc_prop* props;
  const int32_t *svar_order, *sval_order, *soff, *svars;
  orc_itv *store, *root_store, *best_store;
  int store_bot; /* VStore is_bot flag */
  decision_t* dec;
  int32_t dec_cap, depth;
  int32_t cur_strategy, next_unassigned, snap_strategy, snap_next_unassigned;
  int32_t best_bound;
  int stop;
  orc_stats st;
  struct timespec t0;
  uint64_t cur_subproblem;
  int32_t last_obj_ub, dive_left; /* test aid (orc_set_path_sink) */
} engine_t;

static void eng_embed(engine_t* e, int32_t v, int32_t l, int32_t u) {
  int ch = 0, fl = 0;
  embed(e->store, v, l, u, &ch, &fl);
  if (fl) e->store_bot = 1;
}

static inline int splittable(const orc_itv d) { return d.lb != d.ub && !is_inf(d.lb) && !is_inf(d.ub); }

/* VStore::is_extractable<AtomicExtraction> of a store that is not bot: every variable is assigned ("When the problem is extractable, then all
 * variables are assigned to a single value", hybrid_dive_and_solve.hpp:531).  The `gpu` and `cpu` paths ask it of a node whose propagators are all
 * entailed before they call it a solution (gpu_dive_and_solve.hpp:333-338, cpu_solving.hpp:33-40); barebones does not (barebones:988-993). */
static int all_assigned(int32_t n_vars, const orc_itv* store) {
  for (int32_t v = 0; v < n_vars; ++v)
    if (store[v].lb != store[v].ub) return 0;
  return 1;
}
/* is this (non-failed) node a solution leaf? */
static int solution_node(const orc_config* cfg, int32_t n_vars, const orc_itv* store, int32_t n_props, const orc_prop* props) {
  for (int32_t i = 0; i < n_props; ++i)
    if (!orc_ask(&props[i], store)) return 0;
  return !cfg->leaf_requires_assignment || all_assigned(n_vars, store);
}

static void push_decision(engine_t* e, int32_t val_order, int32_t var) {
  if (e->depth + 1 >= e->dec_cap) {
    e->dec_cap *= 2;
    e->dec = (decision_t*)realloc(e->dec, sizeof(decision_t) * (size_t)e->dec_cap);
  }
  decision_t* d = &e->dec[e->depth];
  const orc_itv dom = e->store[var];
  d->var = var;
  d->cur = -1;
  int32_t mid = (int32_t)((int64_t)dom.lb + ((int64_t)dom.ub - (int64_t)dom.lb) / 2);
  switch (val_order) {
    case ORC_VAL_MIN: d->child[0] = (orc_itv){dom.lb, dom.lb}; d->child[1] = (orc_itv){dom.lb + 1, dom.ub}; break;
    case ORC_VAL_MAX: d->child[0] = (orc_itv){dom.ub, dom.ub}; d->child[1] = (orc_itv){dom.lb, dom.ub - 1}; break;
    case ORC_VAL_SPLIT: d->child[0] = (orc_itv){dom.lb, mid}; d->child[1] = (orc_itv){mid + 1, dom.ub}; break;
    default: d->child[0] = (orc_itv){mid + 1, dom.ub}; d->child[1] = (orc_itv){dom.lb, mid}; break;
  }
  d->rope[0] = e->depth + 1;
  d->rope[1] = e->depth > 0 ? e->dec[e->depth - 1].rope[e->dec[e->depth - 1].cur] : -1;
  ++e->depth;
}

/* key to minimise for the lattice_smallest_split orders (barebones:200-219); ties -> lowest index (:322-338) */
static inline int64_t order_key(int32_t var_order, const orc_itv d) {
  switch (var_order) {
    case ORC_FIRST_FAIL: return (int64_t)d.ub - (int64_t)d.lb;
    case ORC_ANTI_FIRST_FAIL: return -((int64_t)d.ub - (int64_t)d.lb);
    case ORC_SMALLEST: return (int64_t)d.lb;
    case ORC_LARGEST: return -(int64_t)d.ub;
    default: return 0;
  }
}

/* returns 1 if a decision was pushed */
static int split(engine_t* e) {
  for (int32_t s = e->cur_strategy; s < e->n_strats; ++s) {
    int32_t n = e->soff[s + 1] - e->soff[s];
    const int32_t* vars = e->svars + e->soff[s];
    int in_store = (n == 0);
    if (in_store) n = e->n_vars;
    int32_t first = n, chosen = n;
    int64_t best = 0;
    for (int32_t i = e->next_unassigned; i < n; ++i) {
      const orc_itv d = e->store[in_store ? i : vars[i]];
      if (!splittable(d)) continue;
      if (first == n) first = i;
      if (e->svar_order[s] == ORC_INPUT_ORDER) { chosen = i; break; }
      int64_t k = order_key(e->svar_order[s], d);
      if (chosen == n || k < best) { best = k; chosen = i; }
    }
    e->next_unassigned = first;
    if (chosen != n) {
      push_decision(e, e->sval_order[s], in_store ? chosen : vars[chosen]);
      return 1;
    }
    e->cur_strategy = s + 1;
    e->next_unassigned = 0;
  }
  return 0;
}

static double elapsed_s(const engine_t* e) {
  struct timespec t1;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - e->t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - e->t0.tv_nsec);
}

/* One node (barebones:903-1031).  Returns 1 if the node is a leaf. */
static int propagate(engine_t* e, int is_dive) {
  uint64_t it = 0, de = 0;
  int failed = fixpoint(e->n_props, e->props, e->store, e->store_bot, &it, &de);
  int leaf = 0;
  if (!failed) {
    if (solution_node(e->cfg, e->n_vars, e->store, e->n_props, e->props)) {
      leaf = 1;
      int accept;
      if (e->obj_var >= 0) {
        accept = e->best_bound > e->store[e->obj_var].lb;
        if (e->cfg->use_fixed_bound && e->store[e->obj_var].lb > e->cfg->fixed_bound) accept = 0; /* dive leaves carry no bound */
      } else accept = 1;
      if (accept) {
        if (e->obj_var >= 0) e->best_bound = e->store[e->obj_var].lb;
        memcpy(e->best_store, e->store, sizeof(orc_itv) * (size_t)e->n_vars);
        if (g_sink && g_sink_count < g_sink_cap) memcpy(g_sink + (size_t)g_sink_count * (size_t)e->n_vars, e->store, sizeof(orc_itv) * (size_t)e->n_vars);
        if (g_sink) g_sink_count++;
        e->st.solutions++;
        e->st.best_subproblem = (int32_t)e->cur_subproblem;
        if (e->cfg->use_fixed_bound) e->stop = 1;
        else if (e->obj_var < 0 && e->cfg->stop_after_n_solutions != 0 && e->st.solutions >= e->cfg->stop_after_n_solutions) {
          e->st.exhaustive = 0; /* common_solving.hpp:858-867 */
          e->stop = 1;
        }
      }
    }
  } else {
    leaf = 1;
  }
  (void)is_dive;
  e->st.fixpoint_iterations += it;
  e->st.num_deductions += de;
  if (g_trace && (int64_t)e->st.nodes < g_trace_cap) g_trace[e->st.nodes] = (unsigned char)(failed ? 1 : 0);
  e->st.nodes++;
  e->st.fails += failed ? 1 : 0;
  if (e->depth > e->st.depth_max) e->st.depth_max = e->depth;
  if (e->cfg->stop_after_n_nodes != 0 && e->st.nodes >= e->cfg->stop_after_n_nodes) { e->st.exhaustive = 0; e->stop = 1; }
  if (e->cfg->timeout_ms != 0 && (e->st.nodes & 255) == 0 && elapsed_s(e) * 1000.0 >= (double)e->cfg->timeout_ms) { e->st.exhaustive = 0; e->stop = 1; }
  return leaf;
}

/* Test aid: replay the path a workgroup of the HIP engine reports when it leaves its kernel (include/turbo_hip.h: tb_session_debug_path) and
 * return the store under its last node.  The replay is the reference's own walk restricted to ONE root-to-node path: root, then the dive along
 * the bits of `subproblem` (barebones:675-714: propagate, split, take child `bit`), then the recorded decisions of the solve loop
 * (barebones:752-864: tell the objective bound that was in force, propagate, split -- which must choose the recorded variable and children --,
 * take the recorded child), then the last node under the last bound.  A node's fixpoint does not depend on the order in which decisions and
 * bounds were added, nor on the snapshot it was restored from, so the result must equal the engine's store bit for bit.
 * (The variable-selection cursors cur_strategy / next_unassigned are carried along the path from the subproblem's root, as after a backtrack,
 * barebones:859-860; they only skip variables that are assigned or unsplittable.)
 * *mismatch_out: -1 the path replays; i >= 0: decision i differs (or node i was a leaf); -2 - l: level l of the dive was a leaf. */
int orc_replay_path(const orc_config* cfg, int32_t n_vars, const orc_itv* root_store, int32_t n_props, const orc_prop* props,
                    int32_t n_strats, const int32_t* strat_var_order, const int32_t* strat_val_order, const int32_t* strat_off, const int32_t* strat_vars,
                    int32_t obj_var, uint64_t subproblem, int32_t dive_levels_left, int32_t n_decisions, const orc_path_decision* decisions,
                    int32_t last_objective_ub, orc_itv* store_out, int32_t* failed_out, int32_t* mismatch_out) {
  engine_t E;
  memset(&E, 0, sizeof(E));
  engine_t* e = &E;
  e->cfg = cfg; e->n_vars = n_vars; e->n_props = n_props; e->n_strats = n_strats; e->obj_var = obj_var;
  e->props = props; e->svar_order = strat_var_order; e->sval_order = strat_val_order; e->soff = strat_off; e->svars = strat_vars;
  size_t sb = sizeof(orc_itv) * (size_t)(n_vars > 0 ? n_vars : 1);
  e->store = (orc_itv*)malloc(sb);
  e->dec_cap = 1024; e->dec = (decision_t*)malloc(sizeof(decision_t) * (size_t)e->dec_cap);
  memcpy(e->store, root_store, sizeof(orc_itv) * (size_t)n_vars);
  for (int32_t v = 0; v < n_vars; ++v) if (e->store[v].lb > e->store[v].ub) e->store_bot = 1;
  int mismatch = -1, failed = 0, done = 0;
  uint64_t it = 0, de = 0;
  int remaining = cfg->subproblems_power;
  /* the dive */
  while (remaining > dive_levels_left && !done) {
    failed = fixpoint(n_props, props, e->store, e->store_bot, &it, &de);
    const int ent = !failed && solution_node(cfg, n_vars, e->store, n_props, props);
    if (failed || ent || !split(e)) { mismatch = -2 - (cfg->subproblems_power - remaining); done = 1; break; }
    --remaining; --e->depth;
    const int bit = (int)((subproblem >> remaining) & 1u);
    eng_embed(e, e->dec[0].var, e->dec[0].child[bit].lb, e->dec[0].child[bit].ub);
  }
  if (!done && dive_levels_left == 0) {
    if (cfg->has_eps_strategy) { if (e->cur_strategy < 1) e->cur_strategy = 1; e->next_unassigned = 0; }
    for (int32_t i = 0; i < n_decisions && !done; ++i) {
      if (obj_var >= 0 && decisions[i].objective_ub != PINF) eng_embed(e, obj_var, NINF, decisions[i].objective_ub);
      failed = fixpoint(n_props, props, e->store, e->store_bot, &it, &de);
      const int ent = !failed && solution_node(cfg, n_vars, e->store, n_props, props);
      if (failed || ent || !split(e)) { mismatch = i; done = 1; break; }
      decision_t* dd = &e->dec[e->depth - 1];
      const orc_path_decision* r = &decisions[i];
      if (dd->var != r->var || dd->child[0].lb != r->children[0].lb || dd->child[0].ub != r->children[0].ub ||
          dd->child[1].lb != r->children[1].lb || dd->child[1].ub != r->children[1].ub || (r->child != 0 && r->child != 1)) { mismatch = i; done = 1; break; }
      dd->cur = r->child;
      eng_embed(e, dd->var, dd->child[dd->cur].lb, dd->child[dd->cur].ub);
    }
    if (!done && obj_var >= 0 && last_objective_ub != PINF) eng_embed(e, obj_var, NINF, last_objective_ub);
  }
  if (!done) failed = fixpoint(n_props, props, e->store, e->store_bot, &it, &de);
  if (store_out) memcpy(store_out, e->store, sizeof(orc_itv) * (size_t)n_vars);
  if (failed_out) *failed_out = failed;
  if (mismatch_out) *mismatch_out = mismatch;
  free(e->store); free(e->dec);
  return 0;
}

int orc_solve(const orc_config* cfg, int32_t n_vars, const orc_itv* root_store,
              int32_t n_props, const orc_prop* props,
              int32_t n_strats, const int32_t* strat_var_order, const int32_t* strat_val_order,
              const int32_t* strat_off, const int32_t* strat_vars,
              int32_t obj_var, orc_itv* best_store_out, int32_t* has_solution_out, orc_stats* stats_out) {
  engine_t E;
  memset(&E, 0, sizeof(E));
  engine_t* e = &E;
  e->cfg = cfg; e->n_vars = n_vars; e->n_props = n_props; e->n_strats = n_strats; e->obj_var = obj_var;
  e->props = props; e->svar_order = strat_var_order; e->sval_order = strat_val_order; e->soff = strat_off; e->svars = strat_vars;
  size_t sb = sizeof(orc_itv) * (size_t)(n_vars > 0 ? n_vars : 1);
  e->store = (orc_itv*)malloc(sb); e->root_store = (orc_itv*)malloc(sb); e->best_store = (orc_itv*)malloc(sb);
  e->dec_cap = 1024; e->dec = (decision_t*)malloc(sizeof(decision_t) * (size_t)e->dec_cap);
  e->best_bound = PINF;
  e->st.exhaustive = 1; e->st.best_subproblem = -1; e->st.best_bound = PINF;
  clock_gettime(CLOCK_MONOTONIC, &e->t0);
  const int d = cfg->subproblems_power;
  const uint64_t num_sub = (uint64_t)1 << d;
  e->st.eps_num_subproblems = num_sub;

  uint64_t idx = 0;
  while (idx < num_sub && !e->stop) {
    e->cur_subproblem = idx;
    /* C. restore the root (barebones:665-672) */
    e->cur_strategy = 0; e->next_unassigned = 0; e->depth = 0;
    memcpy(e->store, root_store, sizeof(orc_itv) * (size_t)n_vars);
    e->store_bot = 0;
    for (int32_t v = 0; v < n_vars; ++v) if (e->store[v].lb > e->store[v].ub) e->store_bot = 1;
    /* D. dive (barebones:675-714): no objective bound is applied while diving (gpu_dive_and_solve.hpp:370-372) */
    int remaining = d, leaf = 0;
    e->last_obj_ub = PINF; e->dive_left = d;
    while (remaining > 0 && !leaf && !e->stop) {
      leaf = propagate(e, 1);
      if (!leaf && !e->stop) { /* (a search stopped by its node budget leaves the store under its last node untouched) */
        if (!split(e)) { leaf = 1; e->st.exhaustive = 0; }
        else {
          --remaining; --e->depth; e->dive_left = remaining;
          int bit = (int)((idx >> remaining) & 1u);
          eng_embed(e, e->dec[0].var, e->dec[0].child[bit].lb, e->dec[0].child[bit].ub);
        }
      }
    }
    uint64_t next_idx = idx + 1;
    if (leaf && !e->stop) {
      /* E. skip the unreachable subtree (barebones:718-741) */
      next_idx = ((idx >> remaining) + 1) << remaining;
      if ((idx & (((uint64_t)1 << remaining) - 1)) == 0) e->st.eps_skipped_subproblems += next_idx - idx;
    } else if (!e->stop) {
      /* F. solve the subproblem (barebones:742-871) */
      if (cfg->has_eps_strategy) { if (e->cur_strategy < 1) e->cur_strategy = 1; e->next_unassigned = 0; }
      while (!e->stop) {
        if (obj_var >= 0) {
          if (cfg->use_fixed_bound) { e->last_obj_ub = cfg->fixed_bound; eng_embed(e, obj_var, NINF, cfg->fixed_bound); }
          else if (e->best_bound != PINF) {
            if (e->best_bound == NINF) { e->stop = 1; break; } /* unbounded objective, barebones:767-770 */
            e->last_obj_ub = e->best_bound - 1;
            eng_embed(e, obj_var, NINF, e->best_bound - 1);
          }
        }
        leaf = propagate(e, 0);
        if (e->stop) break;
        if (!leaf) {
          if (e->depth == 0) { /* snapshot for backtracking (barebones:785-791) */
            memcpy(e->root_store, e->store, sizeof(orc_itv) * (size_t)n_vars);
            e->snap_strategy = e->cur_strategy; e->snap_next_unassigned = e->next_unassigned;
          }
          if (!split(e)) { leaf = 1; e->st.exhaustive = 0; }
          else {
            decision_t* dd = &e->dec[e->depth - 1];
            ++dd->cur;
            dd->obj_ub = e->last_obj_ub;
            eng_embed(e, dd->var, dd->child[dd->cur].lb, dd->child[dd->cur].ub);
          }
        }
        if (leaf) { /* IV. backtrack: rope jump + recompute from the subproblem root (barebones:812-863) */
          if (e->depth == 0) break;
          e->depth = e->dec[e->depth - 1].rope[e->dec[e->depth - 1].cur];
          if (e->depth == -1) break;
          memcpy(e->store, e->root_store, sizeof(orc_itv) * (size_t)n_vars);
          e->store_bot = 0;
          for (int32_t i = 0; i < e->depth - 1; ++i) eng_embed(e, e->dec[i].var, e->dec[i].child[e->dec[i].cur].lb, e->dec[i].child[e->dec[i].cur].ub);
          decision_t* dd = &e->dec[e->depth - 1];
          ++dd->cur;
          eng_embed(e, dd->var, dd->child[dd->cur].lb, dd->child[dd->cur].ub);
          e->cur_strategy = e->snap_strategy; e->next_unassigned = e->snap_next_unassigned;
        }
      }
      if (!e->stop || (cfg->use_fixed_bound && e->st.solutions > 0)) e->st.eps_solved_subproblems += 1;
    }
    idx = next_idx;
  }
  e->st.best_bound = e->best_bound;
  e->st.solve_seconds = elapsed_s(e);
  if (g_last_store) memcpy(g_last_store, e->store, sizeof(orc_itv) * (size_t)n_vars);
  if (g_path_hdr) { /* the path the search stood on when it returned, in the format of orc_replay_path */
    g_path_hdr->subproblem = e->cur_subproblem; g_path_hdr->dive_levels_left = e->dive_left; g_path_hdr->depth = e->depth;
    g_path_hdr->last_objective_ub = e->last_obj_ub; g_path_hdr->decisions = 0;
    for (int32_t i = 0; i < e->depth && i < g_path_cap; ++i) {
      orc_path_decision* r = &g_path_dec[i];
      r->var = e->dec[i].var; r->child = e->dec[i].cur; r->children[0] = e->dec[i].child[0]; r->children[1] = e->dec[i].child[1]; r->objective_ub = e->dec[i].obj_ub;
      g_path_hdr->decisions = i + 1;
    }
  }
  if (has_solution_out) *has_solution_out = e->st.solutions > 0 ? 1 : 0;
  if (best_store_out && e->st.solutions > 0) memcpy(best_store_out, e->best_store, sizeof(orc_itv) * (size_t)n_vars);
  if (stats_out) *stats_out = e->st;
  free(e->store); free(e->root_store); free(e->best_store); free(e->dec);
  return 0;
}
