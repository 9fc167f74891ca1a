import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import orc
kw = eval("dict(%s)" % sys.argv[1])
for i in map(int, sys.argv[2:]):
    d = np.load('/tmp/inst_%d.npz' % i)
    args = dict(gap_rel=1e-2, max_nodes=400, presolve=0, max_pivots=20000); args.update(kw)
    r = orc.solve_milp(d['q'], d['G'], d['h'], d['lb'], d['ub'], d['is_bin'], **args)
    print(i, r['status'], 'obj %.4f lb %.4f rootlp %.4f rootcut %.4f nodes %d piv %d cuts %d' % (r['obj'], r['lower_bound'], r['root_lp'], r['root_bound'], r['nodes'], r['pivots'], r['cuts']))
