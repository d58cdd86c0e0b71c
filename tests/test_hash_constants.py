"""The compact GROUP BY records hash keys inside a window of 2^39 by an invertible multiplication mod 2^39 (rdf_device.h kG2cMul /
kG2cInv, rdf_groupby.hip g2c_*).  What the kernels rely on, checked on the constants the headers carry: the two multipliers are
inverses, and a dense key range lands on the LDS table's slots without first-probe collisions (a wave waits for its unluckiest
lane: with the first multiplier that was tried, 0x9E3779B1 mod 2^32, the expected longest probe chain per 256 look-ups was 27 and
the aggregate pass took 50 ms instead of 2.8)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M39 = (1 << 39) - 1


def _constants():
    text = open(os.path.join(ROOT, "rust_dataframe_amd", "csrc", "rdf_device.h")).read()
    m = re.search(r"kG2cMul\s*=\s*(0x[0-9A-Fa-f]+)ull,\s*kG2cInv\s*=\s*(0x[0-9A-Fa-f]+)ull", text)
    assert m, "kG2cMul / kG2cInv not found in rdf_device.h"
    slots = int(re.search(r"kG2Slots\s*=\s*(\d+)", text).group(1))
    bits = int(re.search(r"kG2PartBits\s*=\s*(\d+)", text).group(1))
    return int(m.group(1), 16), int(m.group(2), 16), slots, bits


def test_compact_hash_is_invertible():
    mul, inv, _, _ = _constants()
    assert mul % 2 == 1 and (mul * inv) & M39 == 1
    rng = np.random.default_rng(5)
    d = rng.integers(0, 1 << 39, 100_000, dtype=np.uint64)
    h = (d * np.uint64(mul)) & np.uint64(M39)
    assert np.array_equal((h * np.uint64(inv)) & np.uint64(M39), d)
    assert len(np.unique(h)) == len(np.unique(d))


def _first_probe_collisions(keys, mul, slots, part_bits):
    h = (keys.astype(np.uint64) * np.uint64(mul)) & np.uint64(M39)
    part = (h >> np.uint64(39 - part_bits)).astype(np.int64)
    rem = h & np.uint64((1 << (39 - part_bits)) - 1)              # the bits below the partition bits pick the slot (tab_upsert, PB > 0)
    frac32 = (rem << np.uint64(32 - (39 - part_bits))).astype(np.uint64)   # as the 32-bit fraction the kernel multiplies by the slot count
    slot = ((frac32 * np.uint64(slots)) >> np.uint64(32)).astype(np.int64)
    worst = 0
    for p in np.unique(part)[:: max(1, (1 << part_bits) // 16)]:  # a sample of the partitions
        s = slot[part == p]
        worst = max(worst, len(s) - len(np.unique(s)))
    return worst


def test_dense_keys_spread_without_collisions():
    mul, _, slots, part_bits = _constants()
    for n, stride, offset in [(100_000, 1, 0), (1_000_000, 1, 12345), (1_300_000, 1, 1 << 38), (1_000_000, 7, 3), (1_000_000, 1000, 0), (300_000, 64, 99)]:
        keys = (np.arange(n, dtype=np.uint64) * np.uint64(stride) + np.uint64(offset)) & np.uint64(M39)
        assert _first_probe_collisions(keys, mul, slots, part_bits) == 0, (n, stride, offset)
    # ... which is not a property of any odd multiplier near the golden ratio: 2^39 / phi rounded to an odd number (continued
    # fraction 1 x 28, then 6, 1, 2, 3, 1, 1, 2, 1, 26, ...) collides on a million consecutive keys
    assert _first_probe_collisions(np.arange(1_000_000, dtype=np.uint64), 0x4F1BBCDCBF, slots, part_bits) > 0
