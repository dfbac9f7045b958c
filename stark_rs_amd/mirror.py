"""Host-side mirror of the reference's types for the hot path -- same names, argument meaning
and error behaviour as 0xSooki/stark-rs, with the heavy methods routed through the C ABI
(include/stark_mi.h) onto the GPU.  The Rust binding a maintainer would add is shown in
INTEGRATION.md; this mirror exists so the parity tests read like the reference's own tests
(tests/test_gpu_mirror.py transliterates them).

Scalar field operations (FiniteField::add/sub/mul/...) are plain host integer arithmetic, as in
the reference (src/ff.rs:138-233) -- they are the glue between kernel calls, not the hot path.
Anything that touches a whole codeword, polynomial or tree goes to the device:

    Polynomial.interpolate_domain / eval_domain / scale   -> smi_intt / smi_coset_ntt / smi_poly_scale
    Hash.from_field_elements / combine / from_bytes       -> smi_hash_* (device kernels, even for one digest)
    MerkleTree.new / commit / open                        -> smi_merkle_*
    FiatShamir.challenge                                   -> smi_hash_bytes
    Fri.fold_codeword / commit / prove                     -> smi_fri_fold / smi_fri_commit / smi_fri_prove

Domains that are not geometric (offset * omega^k) have no fast path: the mirror raises
StarkMiError("domain is not offset*omega^k") -- the Rust shim falls back to the original CPU
code there; this package has no CPU compute path by design.
"""
import numpy as np

from ._lib import StarkMiError
from .engine import P_REF, default_engine

PANIC = StarkMiError  # a reference panic surfaces as this exception with the identical message


class FiniteField:
    """src/ff.rs:9-12,108-233"""

    def __init__(self, p):
        self.p = p

    def __eq__(self, o):
        return isinstance(o, FiniteField) and self.p == o.p

    def __hash__(self):
        return hash(self.p)

    def modulus(self):
        return self.p

    def new_element(self, value):
        return FieldElement(value, self)      # unreduced, like src/ff.rs:113-118

    def zero(self):
        return FieldElement(0, self)

    def one(self):
        return FieldElement(1, self)

    def add(self, l, r):
        return FieldElement((l.value + r.value) % self.p, self)

    def sub(self, l, r):
        return FieldElement((self.p + l.value - r.value) % self.p, self)

    def mul(self, l, r):
        return FieldElement((l.value * r.value) % self.p, self)

    def neg(self, x):
        return FieldElement((self.p - x.value) % self.p, self)

    def inv(self, x):
        if x.value % self.p == 0:
            raise StarkMiError(-1, "no inverse")                  # src/ff.rs:171
        return FieldElement(pow(x.value, -1, self.p), self)

    def div(self, l, r):
        if r.value == 0:
            raise StarkMiError(-2, "no division by zero")         # src/ff.rs:182
        return self.mul(l, self.inv(r))

    def exp(self, base, e):
        return FieldElement(pow(base.value, e, self.p), self)

    def g(self):
        if self.p != P_REF:
            raise StarkMiError(-16, "assertion failed: self.p == 998244353")   # src/ff.rs:192
        return FieldElement(3, self)

    def prim_nth_root(self, n):
        if self.p != P_REF:
            raise StarkMiError(-16, "assertion failed: self.p == 998244353")   # src/ff.rs:216
        if n == 0 or n & (n - 1):
            raise StarkMiError(-3, "n must be a power of two")                 # src/ff.rs:217
        if n > (1 << 23):
            raise StarkMiError(-4, "n > 2^23 not supported by this modulus")   # src/ff.rs:218
        return self.exp(self.g(), (self.p - 1) // n)

    def sample(self, salt: bytes):
        acc = 0
        for b in salt:                                             # src/ff.rs:225-232
            acc = (acc << 8) % self.p
            acc = (acc ^ b) % self.p
        return FieldElement(acc, self)

    def engine(self):
        """The GPU context for this modulus (one per process)."""
        if self.p == P_REF:
            return default_engine()
        from .engine import G2, P2
        if self.p == P2:
            return default_engine(P2, G2)
        raise StarkMiError(-52, f"unsupported modulus for this size: no generator registered for p={self.p}")


class FieldElement:
    """src/ff.rs:24-28 plus the operator impls (:30-106, :235-281)"""
    __slots__ = ("value", "field")

    def __init__(self, value, field):
        self.value, self.field = value, field

    def __eq__(self, o):
        return isinstance(o, FieldElement) and self.value == o.value and self.field == o.field

    def __hash__(self):
        return hash((self.value, self.field.p))

    def __lt__(self, o):
        return self.value < o.value

    def __add__(self, o):
        return self.field.add(self, o)

    def __sub__(self, o):
        return self.field.sub(self, o)

    def __mul__(self, o):
        return self.field.mul(self, o)

    def __truediv__(self, o):
        return self.field.div(self, o)

    def __neg__(self):
        return self.field.neg(self)

    def __xor__(self, e):
        return self.field.exp(self, e)

    def pow(self, e):
        return self.field.exp(self, e)

    def __repr__(self):
        return f"FieldElement({self.value})"


def _vals(elems):
    return np.fromiter((e.value for e in elems), dtype=np.uint64, count=len(elems))


class Polynomial:
    """src/univariate/mod.rs:8-153 (+ interpolate.rs, eval.rs); ascending coefficients."""

    def __init__(self, coeffs, field):
        self.coeffs, self.field = list(coeffs), field

    new = classmethod(lambda cls, coeffs, field: cls(coeffs, field))

    @staticmethod
    def zero_poly(field):
        return Polynomial([], field)

    @staticmethod
    def constant_poly(field, value):
        return Polynomial([field.new_element(value)], field)

    @staticmethod
    def linear_poly(field, a, b):
        return Polynomial([field.new_element(a), field.new_element(b)], field)

    def deg(self):
        d = -1
        for i, c in enumerate(self.coeffs):
            if c.value != 0:
                d = i
        return d

    def is_zero(self):
        return self.deg() == -1

    def __eq__(self, o):                                            # mod.rs:13-39: trailing zeros ignored
        if self.deg() != o.deg():
            return False
        return all(self.coeffs[i] == o.coeffs[i] for i in range(self.deg() + 1))

    def scale(self, factor):
        """f(cX) -- mod.rs:99-113, on the device."""
        if not self.coeffs:
            return Polynomial([], self.field)
        out = self.field.engine().poly_scale(_vals(self.coeffs), factor.value % self.field.p)
        return Polynomial([FieldElement(int(v), self.field) for v in out], self.field)

    def __mul__(self, o):
        """Polynomial::mul (mul.rs:6-29), NTT-based on the device."""
        out = self.field.engine().poly_mul(_vals(self.coeffs), _vals(o.coeffs))
        return Polynomial([FieldElement(int(v), self.field) for v in out], self.field)

    mul = staticmethod(lambda lhs, rhs: lhs * rhs)

    @staticmethod
    def div(numer, denom):
        """Polynomial::div (div.rs:6-42) -> (quotient, remainder), NTT products on the device."""
        q, r = numer.field.engine().poly_div(_vals(numer.coeffs), _vals(denom.coeffs))
        f = numer.field
        return Polynomial([FieldElement(int(v), f) for v in q], f), Polynomial([FieldElement(int(v), f) for v in r], f)

    def __truediv__(self, o):
        return Polynomial.div(self, o)

    @staticmethod
    def intdiv(numer, denom):                                       # div.rs:44-48
        q, r = Polynomial.div(numer, denom)
        assert r.is_zero()
        return q

    @staticmethod
    def modulo(numer, denom):                                       # div.rs:50-53
        return Polynomial.div(numer, denom)[1]

    @staticmethod
    def zerofier(domain):
        """mod.rs:77-96: prod (x - d).  The reference multiplies the factors in one by one; here a
        balanced product tree of device NTT products gives the same polynomial."""
        field = domain[0].field
        polys = [Polynomial([FieldElement((-d.value) % field.p, field), field.one()], field) for d in domain]
        while len(polys) > 1:
            nxt = [polys[i] * polys[i + 1] for i in range(0, len(polys) - 1, 2)]
            if len(polys) % 2:
                nxt.append(polys[-1])
            polys = nxt
        return polys[0]

    @staticmethod
    def interpolate_domain(domain, values):
        """interpolate.rs:6-44 for a geometric domain offset*omega_n^k (the fast-path contract)."""
        if len(domain) != len(values):
            raise StarkMiError(-14, "assertion failed: domain.len() == values.len()")
        if len(domain) == 0:
            raise StarkMiError(-15, "assertion failed: domain.len() > 0")
        field = domain[0].field
        eng = field.engine()
        d = _vals(domain)
        if len(set(int(x) for x in d)) != len(d):
            raise StarkMiError(-1, "no inverse")                    # duplicate points: ff.rs:171 via interpolate.rs:34
        ok, offset = eng.domain_is_geometric(d)
        if not ok:
            raise StarkMiError(-53, "domain is not offset*omega^k")
        v = _vals(values)
        out = eng.intt(v, offset)
        if len(v) > 1 and not v.any():
            return Polynomial([], field)                            # H8: all-zero values give the empty polynomial
        return Polynomial([FieldElement(int(c), field) for c in out], field)

    def eval(self, x):
        """eval.rs:6-14 -- one point: host Horner-equivalent (not the hot path)."""
        p = self.field.p
        acc = 0
        for c in reversed(self.coeffs):
            acc = (acc * x.value + c.value) % p
        return FieldElement(acc, self.field)

    def eval_domain(self, domain):
        """eval.rs:16-21 on a geometric domain, in domain order, on the device."""
        if not domain:
            return []
        field = domain[0].field
        eng = field.engine()
        d = _vals(domain)
        ok, offset = eng.domain_is_geometric(d)
        n_c = self.deg() + 1
        if not ok or n_c > len(d):
            raise StarkMiError(-53, "domain is not offset*omega^k")
        out = eng.coset_ntt(_vals(self.coeffs[:n_c]), len(d).bit_length() - 1, offset)
        return [FieldElement(int(v), field) for v in out]


class Hash:
    """src/hash.rs:1-46; digests are computed by the device kernels."""
    __slots__ = ("bytes",)

    def __init__(self, b):
        self.bytes = bytes(b)

    @property
    def _0(self):
        return self.bytes

    def __eq__(self, o):
        return isinstance(o, Hash) and self.bytes == o.bytes

    def __hash__(self):
        return hash(self.bytes)

    @staticmethod
    def from_bytes(data: bytes):
        return Hash(default_engine().hash_bytes(bytes(data)))

    @staticmethod
    def from_field_elements(elements):
        b = b"".join(int(e).to_bytes(8, "little") for e in elements)    # hash.rs:32-35
        return Hash.from_bytes(b)

    @staticmethod
    def from_u64(value):
        return Hash.from_bytes(int(value).to_bytes(8, "little"))

    @staticmethod
    def combine(left, right):
        return Hash(bytes(default_engine().hash_combine_pairs(np.frombuffer(left.bytes + right.bytes, dtype=np.uint8))[0]))

    def to_hex(self):
        return self.bytes.hex()


class MerkleTree:
    """src/merkle.rs:4-97; the tree lives on the device."""

    def __init__(self, leaves):
        arr = np.frombuffer(b"".join(h.bytes for h in leaves), dtype=np.uint8).reshape(-1, 32) if leaves else np.zeros((0, 32), np.uint8)
        self._t = default_engine().merkle_new(arr)
        self.leaves = list(leaves)
        self.root = Hash(self._t.root())

    new = classmethod(lambda cls, leaves: cls(leaves))

    @property
    def nodes(self):
        """`nodes` of merkle.rs:6: every level, leaves first."""
        lv, n, out = 0, len(self.leaves), []
        while n >= 1:
            out.append([Hash(bytes(r)) for r in self._t.level(lv)])
            lv, n = lv + 1, n // 2
        return out

    def get_root(self):
        return self.root

    @staticmethod
    def commit(leaves):
        arr = np.frombuffer(b"".join(h.bytes for h in leaves), dtype=np.uint8).reshape(-1, 32) if leaves else np.zeros((0, 32), np.uint8)
        return Hash(default_engine().merkle_commit(arr))

    def open(self, index):
        return [Hash(p) for p in self._t.open(index)]

    @staticmethod
    def verify(leaf, index, proof, root):
        cur, idx = leaf, index                                          # merkle.rs:82-96
        for sib in proof:
            cur = Hash.combine(cur, sib) if idx % 2 == 0 else Hash.combine(sib, cur)
            idx //= 2
        return cur == root


class FiatShamir:
    """src/fiat_shamir.rs:4-26"""

    def __init__(self):
        self.transcript = bytearray()

    new = classmethod(lambda cls: cls())

    def absorb(self, data: bytes):
        self.transcript += bytes(data)

    def challenge(self, field):
        h = default_engine().hash_bytes(bytes(self.transcript))
        return field.new_element(int.from_bytes(h[:8], "little"))      # unreduced (H6)


class ProofObject:
    """src/stream.rs:8-14"""
    MERKLE_ROOT, FIELD_ELEMENT, FIELD_ELEMENTS, MERKLE_PATH = 0, 1, 2, 3

    def __init__(self, tag, payload):
        self.tag, self.payload = tag, payload

    def __repr__(self):
        return f"ProofObject({self.tag}, ...)"


class ProofStream:
    """src/stream.rs:4-168 (wire format: tags 0-3, LE u64 lengths/values, raw 32-byte digests)."""

    def __init__(self):
        self.objects = []

    new = classmethod(lambda cls: cls())

    def push(self, obj):
        self.objects.append(obj)

    def pop(self):
        return self.objects.pop(0) if self.objects else None

    def serialize(self) -> bytes:
        out = bytearray()
        for o in self.objects:
            out.append(o.tag)
            if o.tag == 0:
                out += o.payload.bytes
            elif o.tag == 1:
                out += int(o.payload.value).to_bytes(8, "little")
            elif o.tag == 2:
                out += len(o.payload).to_bytes(8, "little")
                for fe in o.payload:
                    out += int(fe.value).to_bytes(8, "little")
            else:
                out += len(o.payload).to_bytes(8, "little")
                for h in o.payload:
                    out += h.bytes
        return bytes(out)

    @staticmethod
    def deserialize(data: bytes, field):
        s, i, n = ProofStream(), 0, len(data)
        while i < n:
            tag = data[i]
            i += 1
            if tag == 0:
                if i + 32 <= n:
                    s.push(ProofObject(0, Hash(data[i:i + 32])))
                    i += 32
            elif tag == 1:
                if i + 8 <= n:
                    s.push(ProofObject(1, field.new_element(int.from_bytes(data[i:i + 8], "little"))))
                    i += 8
            elif tag == 2:
                if i + 8 <= n:
                    cnt = int.from_bytes(data[i:i + 8], "little")
                    i += 8
                    fes = []
                    for _ in range(min(cnt, (n - i) // 8)):
                        fes.append(field.new_element(int.from_bytes(data[i:i + 8], "little")))
                        i += 8
                    s.push(ProofObject(2, fes))
            elif tag == 3:
                if i + 8 <= n:
                    cnt = int.from_bytes(data[i:i + 8], "little")
                    i += 8
                    path = []
                    for _ in range(min(cnt, (n - i) // 32)):
                        path.append(Hash(data[i:i + 32]))
                        i += 32
                    s.push(ProofObject(3, path))
            else:
                break
        return s


class Fri:
    """src/fri.rs:8-311 (prover side).  Verification stays with the reference's CPU code (it is
    tiny and host-side); the tests use the oracle's restatement of Fri::verify."""

    def __init__(self, omega, offset, domain_length, expansion_factor, num_colinearity_tests):
        self.field = omega.field
        self._eng = self.field.engine()
        self._cfg = self._eng.fri_cfg(omega.value, offset.value, domain_length, expansion_factor, num_colinearity_tests)
        self.omega, self.offset = omega, offset
        self.domain_length, self.expansion_factor, self.num_colinearity_tests = domain_length, expansion_factor, num_colinearity_tests

    new = classmethod(lambda cls, *a: cls(*a))

    def num_rounds(self):
        return self._eng.fri_num_rounds(self._cfg)

    def fold_codeword(self, codeword, alpha, offset, omega):
        out = self._eng.fri_fold(_vals(codeword), alpha.value, offset.value, omega.value)
        return [FieldElement(int(v), self.field) for v in out]

    def eval_domain(self, rnd=0):
        p = self.field.p                                               # fri.rs:158-166
        return [FieldElement(self.offset.value * pow(self.omega.value, (1 << rnd) * i, p) % p, self.field)
                for i in range(self.domain_length >> rnd)]

    def commit(self, initial_codeword, proof_stream, fiat_shamir):
        """fri.rs:105-156: pushes the roots and the last codeword, absorbs the roots, and returns
        the codewords of every round (`Vec<Vec<FieldElement>>`), read back from the device."""
        if fiat_shamir.transcript:
            raise StarkMiError(-50, "bad argument: the device transcript starts empty (fresh FiatShamir)")
        roots, alphas, run = self._eng.fri_commit_run(self._cfg, _vals(initial_codeword))
        for r in roots:
            proof_stream.push(ProofObject(0, Hash(bytes(r))))
            fiat_shamir.absorb(bytes(r))
        codewords = [[FieldElement(int(v), self.field) for v in run.codeword(i)] for i in range(len(run))]
        run.free()
        proof_stream.push(ProofObject(2, list(codewords[-1])))
        return codewords

    def prove(self, initial_codeword, fiat_shamir, proof_stream):
        """fri.rs:250-311: fills proof_stream, absorbs the roots into fiat_shamir, returns the
        top-level indices."""
        if fiat_shamir.transcript:
            raise StarkMiError(-50, "bad argument: the device transcript starts empty (fresh FiatShamir)")
        proof, top = self._eng.fri_prove(self._cfg, _vals(initial_codeword))
        for obj in ProofStream.deserialize(proof, self.field).objects:
            proof_stream.push(obj)
            if obj.tag == 0:
                fiat_shamir.absorb(obj.payload.bytes)
        return top


def _fri_verify(self, proof_stream, fiat_shamir, polynomial_values):
    """Fri::verify (src/fri.rs:313-504): host control flow as in the reference; the heavy parts run
    on the device -- leaf hashes and Merkle paths in batches (smi_hash_leaves,
    smi_merkle_verify_batch) and the last-layer interpolation as an inverse NTT instead of the
    reference's O(L^3) Lagrange.  Returns False where the reference prints + returns false."""
    field, p, t = self.field, self.field.p, self.num_colinearity_tests
    eng = self._eng
    omega, offset = self.omega.value, self.offset.value
    R = self.num_rounds()
    roots, alphas = [], []
    for _ in range(R):                                                      # :325-334
        o = proof_stream.pop()
        if o is None or o.tag != ProofObject.MERKLE_ROOT:
            return False
        roots.append(o.payload)
        fiat_shamir.absorb(o.payload.bytes)
        alphas.append(fiat_shamir.challenge(field).value)
    o = proof_stream.pop()                                                   # :337-342
    if o is None or o.tag != ProofObject.FIELD_ELEMENTS:
        return False
    last = [fe.value for fe in o.payload]
    if not roots:
        return False                                                         # "No FRI roots extracted"
    n_last = len(last)
    if n_last == 0 or n_last & (n_last - 1):
        raise StarkMiError(-6, "Number of leaves must be power of 2")       # MerkleTree::new at :353
    if eng.merkle_commit(eng.hash_leaves(last)) != roots[-1].bytes:          # :349-357
        return False
    degree_bound = n_last // self.expansion_factor                            # :360-365
    if degree_bound == 0:
        return False
    last_omega, last_offset = omega, offset
    for _ in range(R - 1):
        last_omega, last_offset = last_omega * last_omega % p, last_offset * last_offset % p
    coeffs = eng.intt(np.array(last, dtype=np.uint64), last_offset) if n_last > 1 else np.array(last, dtype=np.uint64)
    # the iNTT needs last_omega to be the canonical n_last-th root; the round trip below is the
    # reference's own re-evaluation check (:384-390) and fails otherwise
    re_eval = eng.coset_ntt(coeffs, n_last.bit_length() - 1, last_offset)
    if last_omega != eng.prim_nth_root(n_last) or [int(v) for v in re_eval] != last:
        return False
    nz = np.nonzero(coeffs)[0]
    if len(nz) and int(nz[-1]) > degree_bound - 1:                            # :392-397
        return False
    seed = Hash.from_u64(fiat_shamir.challenge(field).value).bytes            # :400-405
    top = self.sample_indices(seed, self.domain_length >> 1, self.domain_length >> (R - 1), t)
    for r in range(R - 1):                                                    # :408-502
        half = self.domain_length >> (r + 1)
        c_idx = [i % half for i in top]
        aa, bb, cc = [], [], []
        for s_ in range(t):
            o = proof_stream.pop()
            if o is None or o.tag != ProofObject.FIELD_ELEMENTS or len(o.payload) != 3:
                return False
            ay, by, cy = (fe.value for fe in o.payload)
            aa.append(ay); bb.append(by); cc.append(cy)
            if r == 0:
                polynomial_values.append((c_idx[s_], field.new_element(ay)))
                polynomial_values.append((c_idx[s_] + half, field.new_element(by)))
            ax = offset * pow(omega, c_idx[s_], p) % p
            bx = offset * pow(omega, c_idx[s_] + half, p) % p
            cx = alphas[r]                                                    # unreduced, like :452
            # test_colinearity (:507-525): (y1-y0)(x2-x0) == (y2-y0)(x1-x0)
            if ((p + by - ay) % p) * ((p + cx - ax) % p) % p != ((p + cy - ay) % p) * ((p + bx - ax) % p) % p:
                return False
        paths = []
        for _ in range(3 * t):
            o = proof_stream.pop()
            if o is None or o.tag != ProofObject.MERKLE_PATH:
                return False
            paths.append(b"".join(h.bytes for h in o.payload))
        for vals, idxs, sel, root in ((aa, c_idx, 0, roots[r]), (bb, [i + half for i in c_idx], 1, roots[r]),
                                      (cc, c_idx, 2, roots[r + 1])):
            mine = [paths[3 * i + sel] for i in range(t)]
            depth = len(mine[0]) // 32 if mine else 0
            if any(len(m_) != depth * 32 for m_ in mine):
                return False
            ok = eng.merkle_verify_batch(eng.hash_leaves(vals), idxs, np.frombuffer(b"".join(mine), dtype=np.uint8), root.bytes)
            if not ok.all():
                return False
        omega, offset = omega * omega % p, offset * offset % p
    return True


def _fri_sample_indices(self, seed, size, reduced_size, number):
    """Fri::sample_indices (src/fri.rs:176-213); digests come from the device hash kernel."""
    if number > 2 * reduced_size:
        raise StarkMiError(-12, "not enough entropy in indices wrt last codeword")
    if number > reduced_size:
        raise StarkMiError(-13, "cannot sample more indices than available in last codeword")
    indices, reduced, counter = [], [], 0
    while len(indices) < number:
        h = Hash.from_bytes(bytes(seed) + counter.to_bytes(4, "little")).bytes
        index = int.from_bytes(h[24:], "big") % size                          # sample_index, :168-174
        counter += 1
        if index % reduced_size not in reduced:
            indices.append(index)
            reduced.append(index % reduced_size)
    return indices


Fri.verify = _fri_verify
Fri.sample_indices = _fri_sample_indices


class Trace:
    """src/trace.rs:3-50"""

    def __init__(self, trace):
        self.trace = [list(r) for r in trace]
        self.num_columns = len(self.trace[0])

    new = classmethod(lambda cls, t: cls(t))

    def get_row(self, i):
        return self.trace[i] if 0 <= i < len(self.trace) else None

    def get_col(self, j):
        return [r[j] for r in self.trace]

    def get(self, i, j):
        try:
            return self.trace[i][j]
        except IndexError:
            return None

    def to_field_elements(self, field):
        return [[field.new_element(e & 0xFFFFFFFFFFFFFFFF) for e in r] for r in self.trace]   # `e as u64`, unreduced

    @staticmethod
    def fibonacci(length):
        a, b, rows = 1, 1, []
        for _ in range(length):
            rows.append([a])
            a, b = b, a + b
        return Trace(rows)

    def lde(self, field, log_blowup, lde_offset=None):
        """Build-defined: column-major low-degree extension of the whole trace on
        the device -- per column interpolate on the trace subgroup, evaluate on the coset."""
        eng = field.engine()
        rows = b"".join(int(v & ((1 << 128) - 1)).to_bytes(16, "little") for r in self.trace for v in r)
        cols = eng.trace_pack(rows, len(self.trace), self.num_columns)
        return eng.lde(cols, log_blowup, 1, lde_offset)


def verify_column_openings(eng, proof: bytes, n_cols, log_n, log_blowup, num_colinearity_tests, column_roots, top_indices):
    """Verifier side of smi_stark_cfg.open_columns (build-defined composition, include/stark_mi.h): the bytes
    after the FRI objects must (1) open every committed column at the layer-0 positions a, b of every
    colinearity test with authentication paths that verify against that column's root (MerkleTree::verify,
    reference src/merkle.rs:82-96, in one device batch per column), and (2) combine, with the Fiat-Shamir
    weights drawn from the column roots (fresh transcript: absorb root c, challenge -- src/fiat_shamir.rs:15-25),
    to the values a and b of the FRI proof's layer-0 triples (src/fri.rs:229-236).  -> bool"""
    p, W, t = eng.p, n_cols, num_colinearity_tests
    N = 1 << (log_n + log_blowup)
    depth, half = log_n + log_blowup, N // 2
    rec, prec = 9 + 8 * W, 9 + 32 * depth
    tail = t * 2 * rec + t * W * 2 * prec
    if len(proof) < tail:
        return False
    fri, ext = proof[:len(proof) - tail], proof[len(proof) - tail:]
    objs = ProofStream.deserialize(fri, FiniteField(p)).objects
    rounds = sum(1 for o in objs if o.tag == 0)
    triples = [o for o in objs if o.tag == 2][1:1 + t]            # after the last codeword: layer 0's t triples
    if len(triples) != t:
        return False
    fs, weights = FiatShamir(), []
    for root in column_roots:
        fs.absorb(bytes(root))
        weights.append(fs.challenge(FiniteField(p)).value % p)
    rows = np.zeros((t, 2, W), dtype=np.uint64)
    for s in range(t):
        for k in range(2):
            r = ext[(2 * s + k) * rec:(2 * s + k + 1) * rec]
            if r[0] != 2 or int.from_bytes(r[1:9], "little") != W:
                return False
            rows[s, k] = np.frombuffer(r[9:], dtype="<u8")
    base = t * 2 * rec
    for c in range(W):
        leaves, idx, paths = [], [], []
        for s in range(t):
            a = top_indices[s] % half
            for k, i in enumerate((a, a + half)):
                rp = ext[base + ((s * W + c) * 2 + k) * prec:base + ((s * W + c) * 2 + k + 1) * prec]
                if rp[0] != 3 or int.from_bytes(rp[1:9], "little") != depth:
                    return False
                leaves.append(int(rows[s, k, c]))
                idx.append(i)
                paths.append(rp[9:])
        digests = eng.hash_leaves(np.array(leaves, dtype=np.uint64))
        if not eng.merkle_verify_batch(digests, idx, np.frombuffer(b"".join(paths), dtype=np.uint8), column_roots[c]).all():
            return False
    for s in range(t):
        for k in range(2):
            acc = sum(weights[c] * int(rows[s, k, c]) for c in range(W)) % p
            if acc != triples[s].payload[k].value % p:
                return False
    return rounds > 0
