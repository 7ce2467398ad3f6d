"""XCSP3-core input (common_solving.hpp:409-413; lala-parsing's parser is absent from the reference tree).

Pinned by the reference's own known answer for this format (benchmarks/test_list.csv: cumulative.xml -> 0) and by
brute-force enumeration of small handwritten instances: the number of solutions of the lowered network (oracle,
all solutions) must equal the number of assignments satisfying the XCSP3 semantics written directly in Python.
"""
import itertools
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import BENCH, ROOT
from oracle import pyoracle
from turbo_amd import capi, frontend

TURBO = os.path.join(ROOT, "turbo_amd", "bin", "turbo")
CUMULATIVE = os.path.join(BENCH, "test_data", "cumulative.xml")


def inst(variables, constraints, objectives=""):
    return (f'<instance format="XCSP3" type="{"COP" if objectives else "CSP"}">\n<variables>{variables}</variables>\n'
            f"<constraints>{constraints}</constraints>\n{objectives}</instance>\n")


def solutions(xml):
    """All solutions of the lowered network projected on the output items (as text), through the oracle."""
    m = frontend.Model.from_xcsp3_string(xml)
    tcn = m.tcn()
    boxes, st = pyoracle.enumerate_solutions(tcn)
    assert st["exhaustive"]
    # a solution leaf is a box (every propagator entailed, some variables possibly free): count the assignments of
    # the instance's own variables it stands for
    own = [v for v in range(tcn.n_vars) if (nm := m.var_name(v)) and not nm.startswith(("X_T", "X_B"))]
    points = set()
    for b in boxes:
        ranges = [range(int(b[v]["lb"]), int(b[v]["ub"]) + 1) for v in own]
        points.update(itertools.product(*ranges))
    return len(points), m, tcn


def optimum(xml):
    m = frontend.Model.from_xcsp3_string(xml)
    tcn = m.tcn()
    has, best, st = pyoracle.solve(tcn)
    assert st["exhaustive"]
    return (tcn.objective_of(best) if has else None), (m.format_solution(best) if has else "")


def test_reference_known_answer_cumulative():
    m = frontend.Model.from_xcsp3_file(CUMULATIVE)
    tcn = m.tcn()
    has, best, st = pyoracle.solve(tcn)
    assert has and st["exhaustive"] and tcn.objective_of(best) == 0
    # check the schedule against the cumulative semantics directly
    starts = [int(v) for v in re.search(r"x = array1d\(0\.\.4, \[([^\]]*)\]\)", m.format_solution(best)).group(1).split(",")]
    lengths, heights = [3, 2, 2, 4, 2], [3, 2, 2, 2, 3]
    for t in range(0, 10):
        assert sum(h for s, l, h in zip(starts, lengths, heights) if s <= t < s + l) <= 5
    assert starts[1] == 0


def test_compact_forms_in_value_lists():
    xml = open(CUMULATIVE).read()
    fzn = frontend.xcsp3_to_fzn(xml)
    assert "array [1..5] of var 0..4: x :: output_array([0..4]);" in fzn
    assert "solve minimize" in fzn
    # heights "3 2x3 3" = 3 2 2 2 3: at t = 0 every task may run
    assert re.search(r"int_lin_le\(\[3, 2, 2, 2, 3\]", fzn)


def count(pred, *domains):
    return sum(1 for t in itertools.product(*domains) if pred(*t))


R5 = range(0, 6)


@pytest.mark.parametrize("name,xml,expected", [
    ("intension_arith", inst('<var id="x"> 0..5 </var><var id="y"> 0..5 </var>',
                             "<intension> eq(add(x,mul(2,y)),7) </intension><intension> lt(x,y) </intension>"),
     count(lambda x, y: x + 2 * y == 7 and x < y, R5, R5)),
    ("intension_logic", inst('<var id="x"> 0..5 </var><var id="y"> 0..5 </var><var id="z"> 0..5 </var>',
                             "<intension> or(and(le(x,1),gt(y,3)),iff(eq(z,2),ne(x,y))) </intension>"
                             "<intension> imp(ge(x,4),eq(dist(y,z),1)) </intension>"),
     count(lambda x, y, z: ((x <= 1 and y > 3) or ((z == 2) == (x != y))) and ((not x >= 4) or abs(y - z) == 1), R5, R5, R5)),
    ("intension_if_minmax", inst('<var id="x"> 0..5 </var><var id="y"> 0..5 </var><var id="z"> 0..5 </var>',
                                 "<intension> eq(z,if(lt(x,y),min(x,3),max(y,2))) </intension><intension> ne(mod(add(x,y),3),neg(sub(0,1))) </intension>"),
     count(lambda x, y, z: z == (min(x, 3) if x < y else max(y, 2)) and (x + y) % 3 != 1, R5, R5, R5)),
    ("alldiff_ordered", inst('<array id="a" size="[4]"> 1..5 </array>',
                             "<allDifferent> a[] </allDifferent><ordered><list> a[0..2] </list><operator> lt </operator></ordered>"),
     count(lambda a, b, c, d: len({a, b, c, d}) == 4 and a < b < c, *[range(1, 6)] * 4)),
    ("sum_coeffs", inst('<array id="a" size="[3]"> 0..4 </array><var id="k"> 3 5 9 </var>',
                        "<sum><list> a[] </list><coeffs> 1 2 3 </coeffs><condition> (eq,k) </condition></sum>"
                        "<sum><list> a[0] a[2] </list><condition> (gt,2) </condition></sum>"),
     count(lambda a, b, c, k: a + 2 * b + 3 * c == k and a + c > 2, range(5), range(5), range(5), [3, 5, 9])),
    ("extension", inst('<var id="x"> 1..3 </var><var id="y"> 0..3 </var><var id="z"> 0..2 </var>',
                       "<extension><list> x y </list><supports> (1,2)(2,3)(*,0) </supports></extension>"
                       "<extension><list> y z </list><conflicts> (0,0)(3,*) </conflicts></extension>"
                       "<extension><list> z </list><supports> 0 2 </supports></extension>"),
     count(lambda x, y, z: ((x, y) in {(1, 2), (2, 3)} or y == 0) and not ((y, z) == (0, 0) or y == 3) and z in (0, 2),
           range(1, 4), range(4), range(3))),
    ("element", inst('<var id="i"> 0..3 </var><var id="v"> 0..9 </var><array id="a" size="[3]"> 0..2 </array><var id="j"> 1..3 </var>',
                     "<element><list> 4 7 1 9 </list><index> i </index><value> v </value></element>"
                     '<element><list startIndex="1"> a[] </list><index> j </index><value> 2 </value></element>'),
     count(lambda i, v, a, b, c, j: [4, 7, 1, 9][i] == v and [a, b, c][j - 1] == 2, range(4), range(10), range(3), range(3), range(3), range(1, 4))),
    ("minmax_group", inst('<array id="m" size="[2][2]"> 0..3 </array>',
                          "<minimum><list> m[][0] </list><condition> (ge,1) </condition></minimum>"
                          "<maximum><list> m[1][] </list><condition> (eq,2) </condition></maximum>"
                          "<group><intension> ne(%0,%1) </intension><args> m[0][0] m[0][1] </args><args> m[0][1] m[1][1] </args></group>"),
     count(lambda a, b, c, d: min(a, c) >= 1 and max(c, d) == 2 and a != b and b != d, *[range(4)] * 4)),
    ("group_variadic", inst('<array id="g" size="[2][3]"> 0..3 </array><var id="t"> 4 5 </var>',
                            "<group><sum><list> %... </list><condition> (eq,%0) </condition></sum><args> t g[0][] </args><args> 5 g[1][] </args></group>"
                            "<group><allDifferent> %... </allDifferent><args> g[0][0] g[1][0] g[1][1] </args></group>"),
     count(lambda a, b, c, d, e, f, t: a + b + c == t and d + e + f == 5 and len({a, d, e}) == 3, *([range(4)] * 6 + [[4, 5]]))),
    ("allequal_count", inst('<array id="a" size="[4]"> 0..2 </array><var id="k"> 0..4 </var><array id="e" size="[2]"> 1..3 </array>',
                            "<allEqual> e[] </allEqual><count><list> a[] </list><values> 0 2 </values><condition> (eq,k) </condition></count>"
                            "<count><list> a[] </list><values> 1 </values><condition> (le,1) </condition></count>"),
     count(lambda a, b, c, d, k, e, f: e == f and sum(v in (0, 2) for v in (a, b, c, d)) == k and sum(v == 1 for v in (a, b, c, d)) <= 1,
           range(3), range(3), range(3), range(3), range(5), range(1, 4), range(1, 4))),
    ("nooverlap", inst('<array id="s" size="[3]"> 0..5 </array>',
                       "<noOverlap><origins> s[] </origins><lengths> 2 3 1 </lengths></noOverlap>"),
     count(lambda a, b, c: all(p + lp <= q or q + lq <= p for (p, lp), (q, lq) in (((a, 2), (b, 3)), ((a, 2), (c, 1)), ((b, 3), (c, 1)))),
           range(6), range(6), range(6))),
    ("channel", inst('<array id="x" size="[3]"> 0..2 </array><array id="y" size="[3]"> 0..2 </array><array id="p" size="[3]"> 1..3 </array>',
                     "<channel><list> x[] </list><list> y[] </list></channel>"
                     '<channel><list startIndex="1"> p[] </list></channel>'),
     count(lambda x0, x1, x2, y0, y1, y2, p1, p2, p3:
           all(((x0, x1, x2)[i] == j) == ((y0, y1, y2)[j] == i) for i in range(3) for j in range(3)) and
           all(((p1, p2, p3)[i] == j + 1) == ((p1, p2, p3)[j] == i + 1) for i in range(3) for j in range(3)),
           *([range(3)] * 6 + [range(1, 4)] * 3))),
    ("instantiation_block", inst('<array id="a" size="[3]"> 0..3 </array>',
                                 "<block><instantiation><list> a[0] </list><values> 2 </values></instantiation>"
                                 "<intension> le(add(a[0],a[1],a[2]),4) </intension></block>"),
     count(lambda a, b, c: a == 2 and a + b + c <= 4, *[range(4)] * 3)),
])
def test_solution_counts_match_brute_force(name, xml, expected):
    n, m, tcn = solutions(xml)
    assert n == expected


def _regular_ok(word):  # the automaton of the test below: a+ b+ over {0 (a), 1 (b)}, ending in state f
    st = "s"
    for c in word:
        st = {("s", 0): "a", ("a", 0): "a", ("a", 1): "f", ("f", 1): "f"}.get((st, c))
        if st is None:
            return False
    return st == "f"


def _circuit_ok(x):  # sub-circuit: x_i = i are loops outside; the others form one cycle
    n = len(x)
    if sorted(x) != list(range(n)):
        return False
    inside = [i for i in range(n) if x[i] != i]
    if not inside:
        return False
    seen, i = set(), inside[0]
    while i not in seen:
        seen.add(i); i = x[i]
    return seen == set(inside)


@pytest.mark.parametrize("name,xml,expected", [
    ("alldiff_except", inst('<array id="a" size="[3]"> 0..3 </array>', "<allDifferent><list> a[] </list><except> 0 </except></allDifferent>"),
     count(lambda a, b, c: all(p != q or p == 0 for p, q in ((a, b), (a, c), (b, c))), *[range(4)] * 3)),
    ("alldiff_lists", inst('<array id="m" size="[2][2]"> 0..1 </array><array id="r" size="[2]"> 0..1 </array>',
                           "<allDifferent><list> m[0][] </list><list> m[1][] </list><list> r[] </list></allDifferent>"),
     count(lambda a, b, c, d, e, f: len({(a, b), (c, d), (e, f)}) == 3, *[range(2)] * 6)),
    ("alldiff_matrix", inst('<array id="m" size="[2][3]"> 0..2 </array>', "<allDifferent><matrix> m[][] </matrix></allDifferent>"),
     count(lambda a, b, c, d, e, f: len({a, b, c}) == 3 and len({d, e, f}) == 3 and a != d and b != e and c != f, *[range(3)] * 6)),
    ("ordered_lengths", inst('<array id="s" size="[3]"> 0..6 </array>', "<ordered><list> s[] </list><lengths> 2 3 </lengths><operator> le </operator></ordered>"),
     count(lambda a, b, c: a + 2 <= b and b + 3 <= c, *[range(7)] * 3)),
    ("lex_lists", inst('<array id="x" size="[2]"> 0..2 </array><array id="y" size="[2]"> 0..2 </array><array id="z" size="[2]"> 0..2 </array>',
                       "<lex><list> x[] </list><list> y[] </list><list> z[] </list><operator> lt </operator></lex>"),
     count(lambda a, b, c, d, e, f: (a, b) < (c, d) < (e, f), *[range(3)] * 6)),
    ("lex_matrix", inst('<array id="m" size="[2][2]"> 0..2 </array>', "<lex><matrix> m[][] </matrix><operator> ge </operator></lex>"),
     count(lambda a, b, c, d: (a, b) >= (c, d) and (a, c) >= (b, d), *[range(3)] * 4)),
    ("sum_var_coeffs_in", inst('<array id="a" size="[2]"> 0..3 </array><array id="w" size="[2]"> 1..2 </array>',
                               "<sum><list> a[] </list><coeffs> w[] </coeffs><condition> (in,3..5) </condition></sum>"
                               "<sum><list> a[0] a[1] </list><condition> (notin,{1,4}) </condition></sum>"),
     count(lambda a, b, u, v: 3 <= a * u + b * v <= 5 and a + b not in (1, 4), range(4), range(4), range(1, 3), range(1, 3))),
    ("nvalues", inst('<array id="a" size="[4]"> 0..3 </array><var id="k"> 1..4 </var>',
                     "<nValues><list> a[] </list><condition> (eq,k) </condition></nValues>"
                     "<nValues><list> a[0..2] </list><except> 0 </except><condition> (le,1) </condition></nValues>"),
     count(lambda a, b, c, d, k: len({a, b, c, d}) == k and len({a, b, c} - {0}) <= 1, *([range(4)] * 4 + [range(1, 5)]))),
    ("cardinality", inst('<array id="a" size="[4]"> 0..3 </array><var id="o"> 0..4 </var>',
                         '<cardinality><list> a[] </list><values closed="true"> 0 1 2 </values><occurs> 1..2 o 1 </occurs></cardinality>'),
     count(lambda a, b, c, d, o: all(v in (0, 1, 2) for v in (a, b, c, d)) and 1 <= [a, b, c, d].count(0) <= 2 and [a, b, c, d].count(1) == o and [a, b, c, d].count(2) == 1,
           *([range(4)] * 4 + [range(5)]))),
    ("element_matrix", inst('<var id="i"> 0..2 </var><var id="j"> 0..2 </var><var id="v"> 0..9 </var>',
                            "<element><matrix> (1,2,3)(4,5,6) </matrix><index> i j </index><value> v </value></element>"),
     count(lambda i, j, v: i < 2 and [[1, 2, 3], [4, 5, 6]][i][j] == v, range(3), range(3), range(10))),
    ("nooverlap_2d", inst('<array id="x" size="[2]"> 0..3 </array><array id="y" size="[2]"> 0..3 </array>',
                          "<noOverlap><origins> (x[0],y[0])(x[1],y[1]) </origins><lengths> (2,1)(1,3) </lengths></noOverlap>"),
     count(lambda a, b, c, d: a + 2 <= b or b + 1 <= a or c + 1 <= d or d + 3 <= c, *[range(4)] * 4)),
    ("cumulative_variable", inst('<array id="s" size="[2]"> 0..3 </array><array id="l" size="[2]"> 1..2 </array><var id="h"> 1..3 </var><array id="e" size="[2]"> 0..5 </array>',
                                 "<cumulative><origins> s[] </origins><lengths> l[] </lengths><ends> e[] </ends><heights> 2 h </heights><condition> (le,3) </condition></cumulative>"),
     count(lambda s0, s1, l0, l1, h, e0, e1: e0 == s0 + l0 and e1 == s1 + l1 and all((2 if s0 <= t < s0 + l0 else 0) + (h if s1 <= t < s1 + l1 else 0) <= 3 for t in range(6)),
           range(4), range(4), range(1, 3), range(1, 3), range(1, 4), range(6), range(6))),
    ("binpacking_condition", inst('<array id="b" size="[3]"> 0..1 </array>', "<binPacking><list> b[] </list><sizes> 2 3 4 </sizes><condition> (le,5) </condition></binPacking>"),
     count(lambda a, b, c: all(sum(sz for sz, bn in zip((2, 3, 4), (a, b, c)) if bn == k) <= 5 for k in (0, 1)), *[range(2)] * 3)),
    ("binpacking_loads", inst('<array id="b" size="[3]"> 0..1 </array><array id="ld" size="[2]"> 0..9 </array>',
                              "<binPacking><list> b[] </list><sizes> 2 3 4 </sizes><loads> ld[] </loads></binPacking>"),
     count(lambda a, b, c, l0, l1: sum(sz for sz, bn in zip((2, 3, 4), (a, b, c)) if bn == 0) == l0 and sum(sz for sz, bn in zip((2, 3, 4), (a, b, c)) if bn == 1) == l1,
           *([range(2)] * 3 + [range(10)] * 2))),
    ("knapsack", inst('<array id="q" size="[3]"> 0..2 </array><var id="p"> 0..20 </var>',
                      "<knapsack><list> q[] </list><weights> 2 3 4 </weights><condition> (le,7) </condition><profits> 3 4 6 </profits><condition> (eq,p) </condition></knapsack>"),
     count(lambda a, b, c, p: 2 * a + 3 * b + 4 * c <= 7 and 3 * a + 4 * b + 6 * c == p, *([range(3)] * 3 + [range(21)]))),
    ("regular", inst('<array id="w" size="[4]"> 0..1 </array>',
                     "<regular><list> w[] </list><transitions> (s,0,a)(a,0,a)(a,1,f)(f,1,f) </transitions><start> s </start><final> f </final></regular>"),
     count(lambda a, b, c, d: _regular_ok((a, b, c, d)), *[range(2)] * 4)),
    ("mdd", inst('<array id="w" size="[3]"> 0..2 </array>',
                 "<mdd><list> w[] </list><transitions> (r,0,n1)(r,1,n2)(n1,2,n3)(n2,0,n3)(n2,1,n4)(n3,1,t)(n4,2,t) </transitions></mdd>"),
     count(lambda a, b, c: (a, b, c) in {(0, 2, 1), (1, 0, 1), (1, 1, 2)}, *[range(3)] * 3)),
    ("clause_slide", inst('<array id="b" size="[4]"> 0..1 </array>',
                          "<clause> b[0] not(b[1]) b[2] </clause><slide><list> b[] </list><intension> le(%0,%1) </intension></slide>"),
     count(lambda a, b, c, d: (a or not b or c) and a <= b <= c <= d, *[range(2)] * 4)),
    ("circuit", inst('<array id="x" size="[4]"> 0..3 </array>', "<circuit> x[] </circuit>"),
     count(lambda a, b, c, d: _circuit_ok((a, b, c, d)), *[range(4)] * 4)),
    ("circuit_size", inst('<array id="x" size="[4]"> 0..3 </array>', "<circuit><list> x[] </list><size> 3 </size></circuit>"),
     count(lambda a, b, c, d: _circuit_ok((a, b, c, d)) and sum(v != i for i, v in enumerate((a, b, c, d))) == 3, *[range(4)] * 4)),
    ("array_cell_domains", inst('<array id="a" size="[3]"><domain for="a[0]"> 1 3 </domain><domain for="others"> 0..1 </domain></array>',
                                "<intension> lt(a[1],a[0]) </intension>"),
     count(lambda a, b, c: b < a, (1, 3), range(2), range(2))),
])
def test_solution_counts_of_the_wider_constraint_set(name, xml, expected):
    n, m, tcn = solutions(xml)
    assert n == expected


def test_sanitizer_corpus_holds_every_instance_above():
    """`make sanitize` runs the ASan + UBSan build of the front end over tests/golden/xcsp3/*.xml: the instances of the two tables above,
    one file each (regenerate a file by writing the `xml` of its row)."""
    rows = []
    for f in (test_solution_counts_match_brute_force, test_solution_counts_of_the_wider_constraint_set):
        for mark in f.pytestmark:
            if mark.name == "parametrize":
                rows += [tuple(getattr(r, "values", r))[:2] for r in mark.args[1]]
    corpus = os.path.join(ROOT, "tests", "golden", "xcsp3")
    assert sorted(os.listdir(corpus)) == sorted(name + ".xml" for name, _ in rows)
    for name, xml in rows:
        assert open(os.path.join(corpus, name + ".xml")).read() == xml, name


def test_objectives():
    v = '<array id="a" size="[3]"> 0..4 </array>'
    c = "<allDifferent> a[] </allDifferent><sum><list> a[] </list><condition> (ge,7) </condition></sum>"
    assert optimum(inst(v, c, "<objectives><minimize> add(a[0],mul(a[1],3)) </minimize></objectives>"))[0] == \
        min(a + 3 * b for a, b, cc in itertools.permutations(range(5), 3) if a + b + cc >= 7)
    assert optimum(inst(v, c, '<objectives><maximize type="sum"><list> a[] </list><coeffs> 1 -2 1 </coeffs></maximize></objectives>'))[0] == \
        max(a - 2 * b + cc for a, b, cc in itertools.permutations(range(5), 3) if a + b + cc >= 7)
    assert optimum(inst(v, c, '<objectives><minimize type="maximum"> a[] </minimize></objectives>'))[0] == \
        min(max(t) for t in itertools.permutations(range(5), 3) if sum(t) >= 7)
    assert optimum(inst(v, c, '<objectives><maximize type="minimum"> a[0] a[1] </maximize></objectives>'))[0] == \
        max(min(t[0], t[1]) for t in itertools.permutations(range(5), 3) if sum(t) >= 7)
    assert optimum(inst(v, c, '<objectives><minimize type="product"> a[0] a[1] </minimize></objectives>'))[0] == \
        min(t[0] * t[1] for t in itertools.permutations(range(5), 3) if sum(t) >= 7)
    c2 = "<sum><list> a[] </list><condition> (ge,7) </condition></sum>"
    assert optimum(inst(v, c2, '<objectives><minimize type="nValues"> a[] </minimize></objectives>'))[0] == \
        min(len(set(t)) for t in itertools.product(range(5), repeat=3) if sum(t) >= 7)


def test_unsupported_and_malformed_inputs_are_reported():
    for bad in ["<instance><variables><var id='x'> 0..2 </var></variables><constraints><precedence> x </precedence></constraints></instance>",
                "<instance><variables><var id='x'> 0..2 </var></variables><constraints><intension> eq(y,1) </intension></constraints></instance>",
                "<instance><variables><var id='x'> 0..2 </variables></instance>",
                "<nothing/>"]:
        with pytest.raises(ValueError, match="XCSP3"):
            frontend.Model.from_xcsp3_string(bad)


def test_cli_rejects_other_extensions(tmp_path):
    p = tmp_path / "model.mzn"
    p.write_text("solve satisfy;\n")
    r = subprocess.run([TURBO, str(p)], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "supported extension: .xml and .fzn" in r.stdout  # config.hpp:268-278


@pytest.mark.gpu
@pytest.mark.parametrize("fixpoint", ["wac1", "event"])
def test_regression_script_row_on_the_gpu(fixpoint):
    # test_turbo.sh:34-44 on the .xml row of test_list.csv
    r = subprocess.run([TURBO, "-eps_var_order", "input_order", "-eps_value_order", "min", "-arch", "barebones", "-s", "-t", "60000",
                        "-fp", fixpoint, CUMULATIVE], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    assert int(re.search(r"objective=(-?\d+)", r.stdout).group(1)) == 0
    assert "==========" in r.stdout and "x = array1d(0..4, [" in r.stdout


@pytest.mark.gpu
def test_xcsp3_network_on_the_engine():
    m = frontend.Model.from_xcsp3_file(CUMULATIVE)
    tcn = m.tcn()
    has, best, st = capi.solve(tcn, capi.make_config(deterministic=1))
    ohas, obest, _ = pyoracle.solve(tcn)
    assert has and ohas and st["exhaustive"]
    np.testing.assert_array_equal(best, obest)  # canonical solution, bit for bit
