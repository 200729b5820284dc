"""Mode-shape selection for TT matrices.

Mirrors the behaviour of the reference's ``t3nsor/utils.py:39-81`` (``auto_shape`` /
``_get_all_factors``) without its sympy / scipy dependencies: n is split into exactly ``d``
integer factors, every candidate is scored, and the best one wins.

Reference semantics restated (not copied):
  * the prime factors of n (with multiplicity) are partitioned into exactly ``d`` non-empty
    groups; when n has fewer than ``d`` prime factors the list is padded with ones first, which
    leaves a single candidate;
  * each candidate is the tuple of group products, ordered by ``mode``;
  * ``criterion='entropy'`` maximises the Shannon entropy of the normalised factors (natural log,
    float64, summed left to right as ``scipy.stats.entropy`` does for short vectors);
    ``criterion='var'`` minimises the variance.
Golden table: tests/golden/g1_auto_shape.npz (generated from the reference, every n <= 4096).
"""
import math

MODES = ['ascending', 'descending', 'mixed']
CRITERIONS = ['entropy', 'var']


def prime_factors(n):
    """Prime factors of n with multiplicity, ascending. prime_factors(1) == []."""
    n = int(n)
    if n < 1:
        raise ValueError("auto_shape needs a positive integer, got {}".format(n))
    out = []
    p = 2
    while p * p <= n:
        while n % p == 0:
            out.append(p)
            n //= p
        p += 1 if p == 2 else 2
    if n > 1:
        out.append(n)
    return out


def _factorizations(n, d, lo=2):
    """All non-decreasing d-tuples of integers >= lo whose product is n."""
    if d == 1:
        if n >= lo:
            yield (n,)
        return
    f = lo
    # the smallest factor f satisfies f**d <= n
    while f ** d <= n:
        if n % f == 0:
            for rest in _factorizations(n // f, d - 1, f):
                yield (f,) + rest
        f += 1


def _interleave(first, last):
    # round-robin merge of the lower and upper halves ('mixed' ordering)
    out = []
    for i in range(max(len(first), len(last))):
        if i < len(first):
            out.append(first[i])
        if i < len(last):
            out.append(last[i])
    return tuple(out)


def _candidates(n, d, mode):
    primes = prime_factors(n)
    if len(primes) < d:
        base = [tuple(sorted(primes + [1] * (d - len(primes))))]
    else:
        base = list(_factorizations(int(n), d))
    if mode == 'ascending':
        return base
    if mode == 'descending':
        return [tuple(reversed(c)) for c in base]
    if mode == 'mixed':
        return [_interleave(c[:len(c) // 2], c[len(c) // 2:]) for c in base]
    raise ValueError('Wrong mode specified, only {} are available'.format(MODES))


def _entropy(factors):
    total = float(sum(factors))
    acc = 0.0
    for f in factors:
        p = float(f) / total
        acc = acc + (-p * math.log(p))
    return acc


def _variance(factors):
    m = sum(factors) / float(len(factors))
    return sum((f - m) ** 2 for f in factors) / float(len(factors))


def auto_shape(n, d=3, criterion='entropy', mode='ascending'):
    """Factor n into d modes (list of ints), picking the most balanced split."""
    cands = _candidates(n, d, mode)
    if criterion == 'entropy':
        score = _entropy
    elif criterion == 'var':
        def score(c):
            return -_variance(c)
    else:
        raise ValueError('Wrong criterion specified, only {} are available'.format(CRITERIONS))
    best, best_score = None, None
    for c in cands:
        s = score(c)
        # ties (never observed for n <= 4096, d <= 4) resolve to the lexicographically largest
        # candidate, i.e. the more balanced leading modes
        if best is None or s > best_score or (s == best_score and c > best):
            best, best_score = c, s
    return [int(v) for v in best]
