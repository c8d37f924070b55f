"""Literal emulation of the reference's GPU radix sort — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

resources/shaders/compute/radix_sort_{upsweep,spine,downsweep}.glsl executed invocation by invocation at the geometry
the shaders are written for: 512-invocation workgroups = 16 subgroups of 32 lanes (radix_sort_spine.glsl:13,
radix_sort_downsweep.glsl:60-61; the index math only works with 32-wide subgroups), 4096-key partitions,
four 8-bit passes ping-ponging between the two halves of the key / value buffers
(gaussian_splatting_rasterizer.gd:144-148: passes 0 and 2 go half 0 -> half 1, passes 1 and 3 back).  Every barrier()
splits the code below into a phase that all invocations of the workgroup finish before the next begins; subgroup
operations act on the active lanes of one 32-lane subgroup; atomicAdd is an order-free integer sum.  Nothing here is
"a stable sort" by construction — the stability of the result is what tests/test_radix_glsl.py and the -m gpu twin test
establish from the shader's own data flow, so that "stable LSD sort on the full 32-bit key, ties in emission order"
(the contract oracle/gsplat_oracle.c and sort.hip implement) is pinned to the shader text, not to prose.
"""
import numpy as np

RADIX = 256
WORKGROUP_SIZE = 512
SUBGROUP = 32
NUM_SUBGROUPS = WORKGROUP_SIZE // SUBGROUP      # 16
PARTITION_DIVISION = 8
PARTITION_SIZE = PARTITION_DIVISION * WORKGROUP_SIZE  # 4096
PAD_KEY = 0xFFFFFFFF


def _sg_exclusive_add(v):
    """subgroupExclusiveAdd over the last axis (32 lanes, all active)."""
    return np.cumsum(v, axis=-1) - v


class Histogram:
    """The `Histogram` buffer of the three shaders (radix_sort_upsweep.glsl:18-22)."""

    def __init__(self, element_count, max_elements):
        self.element_count = int(element_count)
        self.global_histogram = np.zeros(4 * RADIX, np.int64)
        parts = (max_elements + PARTITION_SIZE - 1) // PARTITION_SIZE
        self.partition_histogram = np.zeros(max(parts, 1) * RADIX, np.int64)


def upsweep(h, keys, pas, in_offset):
    """radix_sort_upsweep.glsl:35-65, one workgroup per partition."""
    d = h.element_count
    num_wg = (len(keys) // 2 + PARTITION_SIZE - 1) // PARTITION_SIZE  # the indirect grid only ever covers enough
    for partition_index in range(num_wg):
        partition_start = partition_index * PARTITION_SIZE
        if partition_start >= d:                                     # :45
            continue
        local_histogram = np.zeros(RADIX, np.int64)                  # :47
        index = np.arange(WORKGROUP_SIZE)
        for i in range(PARTITION_DIVISION):                          # :51-56
            key_index = partition_start + WORKGROUP_SIZE * i + index
            key = np.where(key_index < d, keys[np.minimum(key_index, d - 1) + in_offset], PAD_KEY).astype(np.int64)
            radix = (key >> (8 * pas)) & 0xFF
            np.add.at(local_histogram, radix, 1)
        h.partition_histogram[RADIX * partition_index: RADIX * (partition_index + 1)] = local_histogram  # :61
        h.global_histogram[RADIX * pas: RADIX * (pas + 1)] += local_histogram                             # :63


def spine(h, pas):
    """radix_sort_spine.glsl:35-92, workgroup `radix` of 256."""
    d = h.element_count
    partition_count = (d + PARTITION_SIZE - 1) // PARTITION_SIZE
    index = np.arange(WORKGROUP_SIZE)
    sg = index // SUBGROUP
    for radix in range(RADIX):
        reduction = 0                                                # :44
        i = 0
        while WORKGROUP_SIZE * i < partition_count:                  # :47
            partition_index = WORKGROUP_SIZE * i + index
            inside = partition_index < partition_count
            value = np.where(inside, h.partition_histogram[RADIX * np.minimum(partition_index, partition_count - 1)
                                                           + radix], 0)
            v2 = value.reshape(NUM_SUBGROUPS, SUBGROUP)
            excl = (_sg_exclusive_add(v2) + reduction).reshape(-1)   # :50 (reads `reduction` before this trip's update)
            sums = v2.sum(axis=1)                                    # :51
            intermediate = sums.copy()                               # :53
            # :56-63 lanes 0..15 of subgroup 0
            inter_excl = np.cumsum(intermediate) - intermediate
            reduction += int(intermediate.sum())
            intermediate = inter_excl
            excl = excl + intermediate[sg]                           # :67
            w = partition_index[inside]
            h.partition_histogram[RADIX * w + radix] = excl[inside]  # :68
            i += 1
        if radix == 0:                                               # :72-91 global histogram of this pass
            value = h.global_histogram[RADIX * pas: RADIX * (pas + 1)].copy()
            v2 = value.reshape(RADIX // SUBGROUP, SUBGROUP)
            excl = _sg_exclusive_add(v2)
            sums = v2.sum(axis=1)
            inter = np.cumsum(sums) - sums
            h.global_histogram[RADIX * pas: RADIX * (pas + 1)] = (excl + inter[:, None]).reshape(-1)


def downsweep(h, keys, values, pas, in_offset, out_offset):
    """radix_sort_downsweep.glsl:59-214, one workgroup per partition."""
    d = h.element_count
    num_wg = (len(keys) // 2 + PARTITION_SIZE - 1) // PARTITION_SIZE
    sgi = np.arange(NUM_SUBGROUPS)[:, None]      # subgroup_index
    lane = np.arange(SUBGROUP)[None, :]          # thread_index
    index = (sgi * SUBGROUP + lane)              # (16, 32)
    for partition_index in range(num_wg):
        partition_start = partition_index * PARTITION_SIZE
        if partition_start >= d:                                     # :69
            continue
        local_histogram = np.zeros(PARTITION_SIZE, np.int64)         # :71-76 (zeroed where used)
        local_keys = np.zeros((PARTITION_DIVISION, NUM_SUBGROUPS, SUBGROUP), np.int64)
        local_values = np.zeros_like(local_keys)
        local_radix = np.zeros_like(local_keys)
        local_offsets = np.zeros_like(local_keys)
        subgroup_histogram = np.zeros_like(local_keys)
        for i in range(PARTITION_DIVISION):                          # :85-119
            key_index = partition_start + (PARTITION_DIVISION * SUBGROUP) * sgi + i * SUBGROUP + lane
            ok = key_index < d
            src = np.minimum(key_index, max(d - 1, 0)) + in_offset
            key = np.where(ok, keys[src], PAD_KEY).astype(np.int64)
            local_keys[i] = key
            local_values[i] = np.where(ok, values[src], 0)
            radix = (key >> (pas * 8)) & 0xFF
            local_radix[i] = radix
            # :95-102: eight ballots leave, per lane, the set of lanes of its subgroup with the same digit
            same = radix[:, :, None] == radix[:, None, :]            # (sg, lane, other lane)
            lower = np.arange(SUBGROUP)[None, None, :] < np.arange(SUBGROUP)[None, :, None]
            subgroup_offset = (same & lower).sum(axis=2)             # :105
            radix_count = same.sum(axis=2)                           # :106
            elected = subgroup_offset == 0                           # :109
            np.add.at(local_histogram, (NUM_SUBGROUPS * radix + sgi)[elected], radix_count[elected])  # :111
            subgroup_histogram[i] = np.where(elected, radix_count, 0)
            local_offsets[i] = subgroup_offset
        # :122-163 three-level exclusive scan over the 4096 (radix, subgroup) counters, in place
        v = local_histogram.reshape(-1, SUBGROUP)                    # groups of 32 consecutive entries
        sums1 = v.sum(axis=1)                                        # local_histogram_sum[0..127]
        lh = _sg_exclusive_add(v)
        v1 = sums1.reshape(-1, SUBGROUP)                             # :132-140
        sums2 = v1.sum(axis=1)                                       # 4 entries
        s1 = _sg_exclusive_add(v1)
        s2 = np.cumsum(sums2) - sums2                                # :144-148
        s1 = s1 + s2[:, None]                                        # :152-154
        lh = lh + s1.reshape(-1)[:, None]                            # :158-160
        local_histogram = lh.reshape(-1)
        # :165-175 post-scan: ranks inside (radix, subgroup) in iteration order
        for i in range(PARTITION_DIVISION):
            radix = local_radix[i]
            local_offsets[i] = local_offsets[i] + local_histogram[NUM_SUBGROUPS * radix + sgi]
            has = subgroup_histogram[i] > 0
            np.add.at(local_histogram, (NUM_SUBGROUPS * radix + sgi)[has], subgroup_histogram[i][has])
        # :178-181 (local_histogram now holds inclusive sums)
        r = np.arange(RADIX)
        vprev = np.where(r == 0, 0, local_histogram[np.maximum(NUM_SUBGROUPS * r - 1, 0)])
        local_histogram_sum = (h.global_histogram[RADIX * pas + r] + h.partition_histogram[RADIX * partition_index + r]
                               - vprev)
        # :186-188 keys grouped by digit in shared memory
        shared = np.zeros(PARTITION_SIZE, np.int64)
        shared[local_offsets.reshape(-1)] = local_keys.reshape(-1)
        # :192-201 binning (thread `index` handles i = index, index + 512, ...)
        i_all = np.arange(PARTITION_SIZE)
        key = shared[i_all]
        radix = (key >> (pas * 8)) & 0xFF
        dst_offset = local_histogram_sum[radix] + i_all
        wr = dst_offset < d                                          # :196
        keys[dst_offset[wr] + out_offset] = key[wr]
        # :205-213 values follow, unguarded
        shared[local_offsets.reshape(-1)] = local_values.reshape(-1)
        values[dst_offset + out_offset] = shared[i_all]


def sort_pairs(keys_in, values_in, capacity=None):
    """The four passes of gaussian_splatting_rasterizer.gd:144-148 on `count` pairs living in half 0 of buffers of
    2 x capacity words (:79-80).  Returns the sorted (keys, values) from half 0."""
    d = len(keys_in)
    cap = int(capacity if capacity is not None else max(d, 1))
    cap = ((cap + PARTITION_SIZE - 1) // PARTITION_SIZE) * PARTITION_SIZE   # room for the unguarded value writes
    keys = np.zeros(2 * cap + PARTITION_SIZE, np.int64)
    values = np.zeros_like(keys)
    keys[:d] = np.asarray(keys_in, np.uint32)
    values[:d] = np.asarray(values_in, np.uint32)
    h = Histogram(d, cap)
    for pas in range(4):
        in_off, out_off = (0, cap) if pas % 2 == 0 else (cap, 0)
        upsweep(h, keys, pas, in_off)
        spine(h, pas)
        downsweep(h, keys, values, pas, in_off, out_off)
    return keys[:d].astype(np.uint32), values[:d].astype(np.uint32)
