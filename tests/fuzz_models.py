"""Random small FlatZinc models over the constraint vocabulary of the front-end (shared by the fuzz tests)."""
import random

import numpy as np


def random_model(seed: int) -> str:
    rng = random.Random(seed)
    n, nb = rng.randint(2, 5), rng.randint(0, 3)
    lines = []
    for i in range(n):
        lo = rng.randint(-3, 2)
        lines.append(f"var {lo}..{lo + rng.randint(0, 5)}: x{i} :: output_var;")
    for i in range(nb):
        lines.append(f"var bool: b{i} :: output_var;")
    xs, bs = [f"x{i}" for i in range(n)], [f"b{i}" for i in range(nb)]

    def term():
        return rng.choice(xs) if rng.random() < 0.8 else str(rng.randint(-2, 3))

    for _ in range(rng.randint(1, 6)):
        k = rng.random()
        if k < 0.25:
            m = rng.randint(1, 3)
            vs, cs = [rng.choice(xs) for _ in range(m)], [rng.choice([-2, -1, 1, 1, 2]) for _ in range(m)]
            op = rng.choice(["int_lin_le", "int_lin_eq", "int_lin_ne"])
            lines.append(f"constraint {op}([{','.join(map(str, cs))}],[{','.join(vs)}],{rng.randint(-3, 6)});")
        elif k < 0.4:
            lines.append(f"constraint {rng.choice(['int_eq', 'int_le', 'int_ne', 'int_lt'])}({term()},{term()});")
        elif k < 0.55 and bs:
            lines.append(f"constraint {rng.choice(['int_le_reif', 'int_eq_reif', 'int_ne_reif'])}({term()},{term()},{rng.choice(bs)});")
        elif k < 0.65 and bs:
            pos, neg = [rng.choice(bs) for _ in range(rng.randint(0, 2))], [rng.choice(bs) for _ in range(rng.randint(0, 2))]
            if pos or neg:
                lines.append(f"constraint bool_clause([{','.join(pos)}],[{','.join(neg)}]);")
        elif k < 0.75:
            lines.append(f"constraint {rng.choice(['int_plus', 'int_times', 'int_max', 'int_min', 'int_minus', 'int_div', 'int_mod'])}"
                         f"({term()},{term()},{rng.choice(xs)});")
        elif k < 0.85:
            arr = [rng.randint(-2, 4) for _ in range(rng.randint(2, 4))]
            lines.append(f"constraint array_int_element({rng.choice(xs)},[{','.join(map(str, arr))}],{rng.choice(xs)});")
        elif k < 0.92 and bs:
            lines.append(f"constraint bool2int({rng.choice(bs)},{rng.choice(xs)});")
        else:
            lines.append(f"constraint int_abs({rng.choice(xs)},{rng.choice(xs)});")
    g = rng.random()
    goal = f"minimize {rng.choice(xs)}" if g < 0.4 else f"maximize {rng.choice(xs)}" if g < 0.8 else "satisfy"
    # r05: a search annotation over every variable / value order of the reference (barebones_dive_and_solve.hpp:193-221,362-387) on part of the variables,
    # alone or as a seq_search of two (drawn from a generator of its own: the constraints of a seed are what they were)
    ann_rng = random.Random(seed * 7919 + 13)
    ann = ""
    if ann_rng.random() < 0.6:
        def one():
            vs = ann_rng.sample(xs + bs, ann_rng.randint(1, len(xs + bs)))
            return f"int_search([{','.join(vs)}],{ann_rng.choice(VAR_ORDER_NAMES)},{ann_rng.choice(VAL_ORDER_NAMES)},complete)"
        ann = f" :: {one()}" if ann_rng.random() < 0.6 else f" :: seq_search([{one()},{one()}])"
    lines.append(f"solve{ann} {goal};")
    return "\n".join(lines) + "\n"


VAR_ORDER_NAMES = ["input_order", "first_fail", "anti_first_fail", "smallest", "largest"]
VAL_ORDER_NAMES = ["indomain_min", "indomain_max", "indomain_split", "indomain_reverse_split"]


NINF, PINF = -2**31, 2**31 - 1


def random_network(rng):
    """Random ternary network (numpy store, props) over all operators with small, Boolean, wide, huge and unbounded domains."""
    from turbo_amd.frontend import ITV_DTYPE, PROP_DTYPE
    V = int(rng.integers(6, 25))
    store = np.zeros(V, dtype=ITV_DTYPE)
    store[0], store[1], store[2] = (0, 0), (1, 1), (2, 2)
    for v in range(3, V):
        kind = rng.random()
        if kind < 0.35:      # small
            lo = int(rng.integers(-6, 6)); hi = lo + int(rng.integers(0, 8))
        elif kind < 0.55:    # Boolean
            lo, hi = 0, 1
        elif kind < 0.8:     # wide
            lo = int(rng.integers(-10**6, 10**6)); hi = lo + int(rng.integers(0, 10**6))
        elif kind < 0.9:     # huge (saturation)
            lo = int(rng.integers(-2**30, 2**30)); hi = min(PINF - 1, lo + int(rng.integers(0, 2**30)))
        else:                # half or fully unbounded
            lo = NINF if rng.random() < 0.7 else int(rng.integers(-100, 100))
            hi = PINF if rng.random() < 0.7 else max(lo if lo != NINF else -100, int(rng.integers(-100, 100)))
        store[v] = (lo, hi)
    P = int(rng.integers(4, 41))
    props = np.zeros(P, dtype=PROP_DTYPE)
    for i in range(P):
        op = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 6, 6, 7, 7]))
        x = int(rng.integers(0, V)) if op < 6 else int(rng.choice([0, 1, int(rng.integers(0, V))]))
        props[i] = (op, x, int(rng.integers(0, V)), int(rng.integers(0, V)))
    return store, props




def channelling_network(rng):
    """Random network made of what the event kernel evaluates jointly: channelling propagators b = (y = k) of a few integer
    variables y over runs of consecutive constants, runs with gaps, duplicated constants and truth variables shared between
    two y, plus implications b1 <= b2 and sums over the Booleans (several readers per Boolean and per y, so that the
    successor slots overflow for one bound event and not the other, and y's readers are dealt over its group's lanes).
    More than 64 records per class, so that slices are class-pure and hold one to three groups."""
    from turbo_amd.frontend import ITV_DTYPE, PROP_DTYPE
    store = [(0, 0), (1, 1), (2, 2)]
    const_of = {0: 0, 1: 1, 2: 2}

    def const(k):
        if k not in const_of:
            const_of[k] = len(store)
            store.append((k, k))
        return const_of[k]
    props = []
    bools = []
    n_y = int(rng.integers(4, 10))
    for _ in range(n_y):
        lo = int(rng.integers(-5, 6))
        width = int(rng.integers(3, 70))
        y = len(store)
        store.append((lo, lo + width))
        ks = list(range(lo - int(rng.integers(0, 3)), lo + width + 1 + int(rng.integers(0, 3))))
        style = rng.random()
        if style < 0.25:    # gaps
            ks = [k for k in ks if rng.random() < 0.7]
        elif style < 0.4:   # a duplicate
            ks.insert(int(rng.integers(0, len(ks))), int(rng.choice(ks)))
        for k in ks:
            if bools and rng.random() < 0.05:
                b = int(rng.choice(bools))  # a truth variable shared by two rules
            else:
                b = len(store)
                store.append((0, 1))
                bools.append(b)
            props.append((6, b, y, const(k)))  # b = (y = k)
        if rng.random() < 0.5:  # a second reader of y: b' = (y <= k)
            b = len(store); store.append((0, 1)); bools.append(b)
            props.append((7, b, y, const(int(rng.integers(lo, lo + width + 1)))))
    for _ in range(int(rng.integers(20, 120))):  # implications
        a, c = int(rng.choice(bools)), int(rng.choice(bools))
        props.append((7, 1, a, c))
    for _ in range(int(rng.integers(0, 12))):  # t = a + c chains
        a, c = int(rng.choice(bools)), int(rng.choice(bools))
        t = len(store); store.append((0, 2))
        props.append((0, t, a, c))
        if rng.random() < 0.5:
            props.append((7, 1, t, 1))  # at most one of the two
    order = rng.permutation(len(props))
    st = np.array(store, dtype=ITV_DTYPE)
    pr = np.array([props[i] for i in order], dtype=PROP_DTYPE)
    return st, pr


def finite_class_network(rng, scale=1):
    """Random SATISFIABLE network for the lean class runs of the event kernels (kernels.hpp: lean_class_run): a hidden solution, and a
    few classes of constraints that hold in it, each with more than 64 records -- so that the engine's class sort (and padding) makes
    class-pure slices -- over small finite integer domains, Booleans and constants: sums x = y + z (also with a constant sum,
    `0 = y + z`, and with Boolean terms), min / max (also the clause `1 = max(b1, b2)`), y <= z, y > z, y = z, y != z, reified
    comparisons between two variables and against constants, implications between Booleans.  Records share variables (defined
    variables feed later records), so successor slots overflow and slices re-run each other.
    `scale` stretches the integer domains (1: widths up to 40; 8: widths on both sides of 255; 400: values on both sides of +-16383),
    for the layouts that pack an integer by its width (COMPACT8)."""
    from turbo_amd.frontend import ITV_DTYPE, PROP_DTYPE
    store = [(0, 0), (1, 1), (2, 2)]
    val = [0, 1, 2]
    const_of = {0: 0, 1: 1, 2: 2}

    def new_var(lo, hi, v):
        store.append((lo, hi)); val.append(v)
        return len(store) - 1

    def const(k):
        if k not in const_of:
            const_of[k] = new_var(k, k, k)
        return const_of[k]
    ints, bools = [], []
    for _ in range(int(rng.integers(12, 40))):
        lo = int(rng.integers(-20 * scale, 30 * scale)); hi = lo + int(rng.integers(0, 40 * min(scale, 8)))
        ints.append(new_var(lo, hi, int(rng.integers(lo, hi + 1))))
    for _ in range(int(rng.integers(10, 40))):
        bools.append(new_var(0, 1, int(rng.integers(0, 2))))

    def iv():
        return int(rng.choice(ints))

    def bv():
        return int(rng.choice(bools))

    def anyv():
        r = rng.random()
        return iv() if r < 0.6 else (bv() if r < 0.9 else const(int(rng.integers(-3, 8))))

    def defined(v):  # a fresh integer variable whose hidden value is v
        x = new_var(v - int(rng.integers(0, 12 * min(scale, 8))), v + int(rng.integers(0, 12 * min(scale, 8))), v)
        ints.append(x)
        return x

    def truth(t):  # a fresh Boolean whose hidden value is t (sometimes an existing one that happens to agree)
        if rng.random() < 0.2:
            for _ in range(4):
                b = bv()
                if val[b] == int(t):
                    return b
        b = new_var(0, 1, int(t)); bools.append(b)
        return b
    props = []
    kinds = rng.choice(["add", "add0", "addb", "minmax", "clause", "leq", "eq", "neq", "gt", "leq_r", "eq_r", "leq_rc", "impl", "mul"], size=int(rng.integers(2, 6)), replace=False)
    for kind in kinds:
        for _ in range(int(rng.integers(66, 150))):
            if kind == "add":
                y, z = anyv(), anyv(); props.append((0, defined(val[y] + val[z]), y, z))
            elif kind == "mul":
                # x = y * z over non-negative operands (r05: the lean product run, kernels.hpp K_MUL_NN): a price times a 0/1 occupancy as in wordpress7_500, two small
                # factors, a constant factor; now and then a factor that may be negative, which sends the whole slice back to the generic rule
                r = rng.random()
                vy = int(rng.integers(0, 30 * min(scale, 8)))
                y = new_var(max(0, vy - int(rng.integers(0, 9))) if rng.random() < 0.93 else vy - 40, vy + int(rng.integers(0, 9)), vy); ints.append(y)
                if r < 0.4: z = bv()
                elif r < 0.8:
                    vz = int(rng.integers(0, 7)); z = new_var(max(0, vz - int(rng.integers(0, 3))), vz + int(rng.integers(0, 3)), vz); ints.append(z)
                else: z = const(int(rng.integers(0, 5)))
                v = val[y] * val[z]
                x = new_var(max(0, v - int(rng.integers(0, 12))), v + int(rng.integers(0, 12)), v); ints.append(x)
                props.append((1, x, y, z))
            elif kind == "add0":
                y = iv(); props.append((0, 0, y, defined(-val[y])))            # 0 = y + z
            elif kind == "addb":
                y, z = bv(), bv(); props.append((0, defined(val[y] + val[z]), y, z))  # i = b1 + b2
            elif kind == "minmax":
                y, z = anyv(), anyv(); mx = int(rng.random() < 0.5)
                props.append((5 if mx else 4, defined(max(val[y], val[z]) if mx else min(val[y], val[z])), y, z))
            elif kind == "clause":
                y, z = bv(), bv()
                if val[y] or val[z]: props.append((5, 1, y, z))                 # 1 = max(b1, b2)
            elif kind in ("leq", "gt", "impl"):
                y, z = (bv(), bv()) if kind == "impl" else (anyv(), anyv())
                if kind == "gt":
                    if val[y] == val[z]: continue
                    if val[y] < val[z]: y, z = z, y
                    props.append((7, 0, y, z))
                else:
                    if val[y] > val[z]: y, z = z, y
                    props.append((7, 1, y, z))
            elif kind == "eq":
                y = anyv(); props.append((6, 1, y, defined(val[y])))
            elif kind == "neq":
                y, z = anyv(), anyv()
                if val[y] != val[z]: props.append((6, 0, y, z))
            elif kind == "leq_r":
                y, z = iv(), iv(); props.append((7, truth(val[y] <= val[z]), y, z))
            elif kind == "eq_r":
                y, z = iv(), anyv(); props.append((6, truth(val[y] == val[z]), y, z))
            else:  # leq_rc
                y = iv(); k = int(rng.integers(-20 * scale, 60 * scale)); props.append((7, truth(val[y] <= k), y, const(k)))
    order = rng.permutation(len(props))
    st = np.array(store, dtype=ITV_DTYPE)
    pr = np.array([props[i] for i in order], dtype=PROP_DTYPE)
    return st, pr


def element_model(seed: int) -> str:
    """FlatZinc model shaped like the headline instance (wordpress7_500): a few index variables over 70-260 positions, each read by two
    to four `array_int_element` constraints with tables over few distinct values, linear constraints over the looked-up values and an
    objective.  Lowered, an index becomes a CHAIN of channelling propagators `b_i = (idx = i)` over several 64-record slices, a value one
    over its table's values, linked by implications `b_i <= c_table[i]` -- what the r04 wake-up filters work on (chain slices woken by
    value range; a chain woken by "b_i became false" only when i sits on a bound of idx)."""
    rng = random.Random(seed)
    n_idx = rng.randint(2, 3)
    lines, idxs, vals = [], [], []
    tables = 0
    for a in range(n_idx):
        n = rng.randint(70, 260)
        lo = rng.choice([1, 1, 1, 0, 3])  # (an index domain that starts elsewhere than the table: the front-end clips it)
        lines.append(f"var {lo}..{n + rng.choice([0, 0, 2])}: i{a} :: output_var;")
        idxs.append(f"i{a}")
        for _ in range(rng.randint(2, 4)):
            distinct = rng.randint(3, 40)
            base = rng.randint(-5, 20)
            style = rng.random()
            if style < 0.4:      # sorted table (offers by size): a value bound cuts a contiguous range of positions
                tab = sorted(base + rng.randint(0, distinct) * rng.choice([1, 1, 2]) for _ in range(n))
            elif style < 0.6:
                tab = sorted((base + rng.randint(0, distinct) for _ in range(n)), reverse=True)
            else:
                tab = [base + rng.randint(0, distinct) for _ in range(n)]
            lines.insert(0, f"array [1..{n}] of int: t{tables} = [{','.join(map(str, tab))}];")
            lines.append(f"var {min(tab) - rng.choice([0, 0, 2])}..{max(tab) + rng.choice([0, 0, 3])}: v{tables} :: output_var;")
            lines.append(f"constraint array_int_element(i{a},t{tables},v{tables});")
            vals.append((f"v{tables}", min(tab), max(tab)))
            tables += 1
    for _ in range(rng.randint(2, 6)):
        m = rng.randint(2, min(4, len(vals)))
        pick = rng.sample(vals, m)
        cs = [rng.choice([-2, -1, 1, 1, 2]) for _ in range(m)]
        mid = sum(c * (lo + hi) // 2 for c, (_, lo, hi) in zip(cs, pick))
        lines.append(f"constraint int_lin_le([{','.join(map(str, cs))}],[{','.join(v for v, _, _ in pick)}],{mid + rng.randint(-6, 10)});")
    if rng.random() < 0.5 and len(idxs) > 1:
        lines.append(f"constraint int_lin_le([1,-1],[{idxs[0]},{idxs[1]}],{rng.randint(-20, 40)});")
    obj = rng.sample(vals, min(len(vals), rng.randint(2, 4)))
    lo, hi = sum(l for _, l, _ in obj), sum(h for _, _, h in obj)
    lines.append(f"var {lo}..{hi}: obj :: output_var;")
    lines.append(f"constraint int_lin_eq([{','.join(['1'] * len(obj))},-1],[{','.join(v for v, _, _ in obj)},obj],0);")
    order = rng.choice(["first_fail", "input_order", "smallest"])
    val = rng.choice(["indomain_min", "indomain_split", "indomain_max"])
    goal = rng.choice(['minimize', 'minimize', 'maximize'])
    # r05: every second model searches with the orders r04 never drew -- anti_first_fail, largest, indomain_reverse_split -- and a seq_search that also
    # branches on the looked-up values (a generator of its own: the constraints of a seed are what they were)
    ann_rng = random.Random(seed * 104729 + 7)
    if ann_rng.random() < 0.5:
        a = f"int_search([{','.join(idxs)}],{ann_rng.choice(VAR_ORDER_NAMES)},{ann_rng.choice(VAL_ORDER_NAMES)},complete)"
        b = f"int_search([{','.join(v for v, _, _ in vals)}],{ann_rng.choice(['anti_first_fail', 'largest', 'smallest'])},{ann_rng.choice(VAL_ORDER_NAMES)},complete)"
        ann = a if ann_rng.random() < 0.5 else f"seq_search([{b},{a}])" if ann_rng.random() < 0.5 else f"seq_search([{a},{b}])"
        lines.append(f"solve :: {ann} {goal} obj;")
    else:
        lines.append(f"solve :: int_search([{','.join(idxs)}],{order},{val},complete) {goal} obj;")
    return "\n".join(lines) + "\n"
