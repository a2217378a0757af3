"""ORACLE (test infrastructure, never imported by the product path): the ORDER in which ATen's CPU kernel adds the
elements of a contiguous fp32 vector -- the one step of the reference's MSEFast loss that is not arithmetic of the
algorithm but of the machine it ran on.

The reference's loss is ``(pred - tgt).abs().pow(2).mean()`` (quantization/observer.py:420-432) on CPU fp32 tensors.
``mean`` on the CPU is ``sum_out(...).div_(n)`` (aten/src/ATen/native/ReduceOps.cpp, mean_out), and ``sum`` of a contiguous
fp32 vector is ``cascade_sum`` -> ``vectorized_inner_sum`` (aten/src/ATen/native/cpu/SumKernel.cpp; PyTorch 2.x, restated
here from its published source -- torch is a third-party dependency of the reference, not vendored under /root/reference;
SURVEY.md 8c lists it as unpinned):

  * the vector is read as n // W SIMD vectors of W fp32 lanes (W = 16 with AVX-512, 8 with AVX2), so lane l only ever
    meets elements l, l + W, l + 2W, ...;
  * ``row_sum``: the vectors are dealt to 4 interleaved accumulators (ILP), vector i going to accumulator i % 4, for the
    first 4 * (n_vec // 4) vectors; each accumulator is a 4-level cascade (``multi_row_sum``): level 0 absorbs 2^p
    consecutive additions (p = max(4, ceil(log2(len)) // 4)), is then flushed into level 1 and zeroed, level 1 into level
    2 when the index is a multiple of 2^(2p), and so on; at the end the levels are folded into level 0 in order;
    left-over vectors go to accumulator 0, then accumulators 1, 2, 3 are added to accumulator 0 in that order;
  * the n % W trailing scalars are added one by one to a scalar that starts at 0, and the W lanes of the vector
    accumulator are then added to it in lane order.

Every addition is an fp32 addition.  Valid for the serial path: fewer than 32768 elements (at::internal::GRAIN_SIZE), or a
one-thread pool -- beyond that ``parallel_reduce`` splits the vector by the machine's thread count.  Rows of weight
matrices (768 / 3072 columns: configs[3]'s per-channel searches) are always serial.

float64 (a per-tensor observer's second call on runs on a float64 copy of x, observer.py:524,549): the same algorithm on
vectors of FOUR doubles (``aten_sum(x, 4, np.float64)``); with a one-thread pool (torch.set_num_threads(1), how
tests/golden/make_golden.py runs the reference) the serial order also holds at and beyond 32768 elements
(``serial_only=False``).

Pinned by tests/test_oracle_pinning.py::test_aten_sum_order against torch.sum itself on whatever machine runs the test,
and by tests/test_oracle_golden.py against the reference-generated rows of tests/golden/msefast_rows.npz (generated on an
AVX-512 machine: W = 16)."""
import numpy as np

F32 = np.float32
SERIAL_LIMIT = 32768          # at::internal::GRAIN_SIZE


def _ceil_log2(x):
    return 0 if x <= 1 else int(x - 1).bit_length()


def _multi_row_sum(rows, nrows, dtype=F32):
    """multi_row_sum<acc_t, nrows>: ``rows`` is [size, nrows, ...] fp32; returns [nrows, ...]: each of the nrows columns
    summed over `size` with the 4-level cascade.  Trailing dimensions (SIMD lanes, batch) are element-wise."""
    size = rows.shape[0]
    num_levels = 4
    level_power = max(4, _ceil_log2(size) // num_levels)
    level_step = 1 << level_power
    level_mask = level_step - 1
    acc = np.zeros((num_levels,) + rows.shape[1:], dtype=dtype)
    i = 0
    while i + level_step <= size:
        for _ in range(level_step):
            acc[0] = acc[0] + rows[i]
            i += 1
        for j in range(1, num_levels):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = 0
            mask = level_mask << (j * level_power)
            if (i & mask) != 0:
                break
    while i < size:
        acc[0] = acc[0] + rows[i]
        i += 1
    for j in range(1, num_levels):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def aten_sum(x, vec, dtype, serial_only=True):
    """torch.sum over the LAST axis of contiguous data of shape [..., n], as ATen's CPU kernel adds it in ``dtype``
    (np.float32 / np.float64: every addition is one of that type) on a machine whose SIMD vectors hold ``vec`` elements,
    on the serial path.  Leading axes are independent problems (batched here only for speed)."""
    F32 = dtype                                     # noqa: N806 -- the body below is written in terms of "the" float type
    x = np.ascontiguousarray(x, dtype=dtype)
    n = x.shape[-1]
    lead = x.shape[:-1]
    if serial_only and n >= SERIAL_LIMIT:
        raise ValueError("aten_sum restates the serial path only (fewer than 32768 elements, or a one-thread pool)")
    if n < vec:
        # size0 < Vec::size(): scalar_inner_sum -> the same cascade on scalars, 4-way ILP
        return _scalar_inner_sum(x, dtype)
    n_vec = n // vec
    v = np.moveaxis(x[..., :n_vec * vec].reshape(lead + (n_vec, vec)), -2, 0)           # [n_vec, ..., vec]
    ilp = 4
    size_ilp = n_vec // ilp
    grouped = np.moveaxis(v[:size_ilp * ilp].reshape((size_ilp, ilp) + v.shape[1:]), 1, 1)   # [size_ilp, ilp, ..., vec]
    partial = _multi_row_sum(grouped, ilp, dtype)                                            # [ilp, ..., vec]
    for i in range(size_ilp * ilp, n_vec):
        partial[0] = partial[0] + v[i]
    for k in range(1, ilp):
        partial[0] = partial[0] + partial[k]
    lanes = partial[0]                                                                        # [..., vec]
    final = np.zeros(lead, dtype=F32)
    for k in range(n_vec * vec, n):
        final = (final + x[..., k]).astype(F32)
    for lane in range(vec):
        final = (final + lanes[..., lane]).astype(F32)
    return final


def aten_sum_flat(x, vec, dtype):
    """``aten_sum(x, vec, dtype, serial_only=False)`` for ONE flat vector of any length, with the cascade's levels walked
    block-wise: every level-0 block (2^p rows), level-1 chunk (2^p blocks) and level-2 unit (2^p chunks) is an independent
    left-to-right sum that starts from zero, so all blocks of a level are added side by side (numpy over the block axis)
    -- the additions and their order are those of ``_multi_row_sum``, which walks the rows one by one (pinned against it
    and against torch.sum on one thread up to 40 M elements, tests/test_oracle_pinning.py).  This is what makes the
    reference's order affordable at the per-tensor sites of configs[3] ([32,128,768] ... [32,128,3072], millions of
    elements per loss evaluation); the device kernels (csrc/aten_order.h) decompose the sum the same way."""
    x = np.ascontiguousarray(x, dtype=dtype).ravel()
    n = x.size
    if n < vec:
        return _scalar_inner_sum(x, dtype)
    n_vec = n // vec
    nc = 4 * vec                                  # 4 interleaved accumulators x vec lanes = independent columns
    size = n_vec // 4                             # rows every column's cascade sees
    power = max(4, _ceil_log2(size) // 4)
    step = 1 << power
    rows = x[:size * nc].reshape(size, nc)

    def seq(a):                                   # [groups, len, nc] -> [groups, nc]: 0 + a[:, 0] + a[:, 1] + ... in order
        acc = np.zeros((a.shape[0], a.shape[2]), dtype=dtype)
        for j in range(a.shape[1]):
            acc = acc + a[:, j]
        return acc

    def chain(a):                                 # [len, nc] -> [nc]
        return seq(a[None])[0]

    blocks = size // step
    lv0 = seq(rows[:blocks * step].reshape(blocks, step, nc))
    chunks = blocks // step
    lv1 = seq(lv0[:chunks * step].reshape(chunks, step, nc))
    units = chunks // step
    lv2 = seq(lv1[:units * step].reshape(units, step, nc))
    acc3 = chain(lv2)
    acc2 = chain(lv1[units * step:])              # what the levels still hold when the rows run out
    acc1 = chain(lv0[chunks * step:])
    acc0 = chain(rows[blocks * step:])
    col = (((acc0 + acc1) + acc2) + acc3).reshape(4, vec)
    for v in range(size * 4, n_vec):              # left-over vectors join accumulator 0
        col[0] = col[0] + x[v * vec:(v + 1) * vec]
    lanes = ((col[0] + col[1]) + col[2]) + col[3]
    final = dtype(0)
    for k in range(n_vec * vec, n):
        final = dtype(final + x[k])
    for lane in range(vec):
        final = dtype(final + lanes[lane])
    return final


def aten_mean_flat(x, vec, dtype):
    """torch.mean of a flat tensor of any length on one thread: sum_out(...).div_(n) in ``dtype``."""
    x = np.asarray(x)
    return dtype(aten_sum_flat(x, vec, dtype) / dtype(x.size))


def aten_sum_f32(x, vec=16):
    """fp32 rows shorter than 32768 elements (the per-channel searches)."""
    return aten_sum(x, vec, np.float32)


def _scalar_inner_sum(x, dtype=F32):
    """scalar_inner_sum (row shorter than one SIMD vector): row_sum on scalars."""
    n = x.shape[-1]
    v = np.moveaxis(x, -1, 0)                                   # [n, ...]
    ilp = 4
    size_ilp = n // ilp
    grouped = v[:size_ilp * ilp].reshape((size_ilp, ilp) + v.shape[1:])
    partial = _multi_row_sum(grouped, ilp, dtype)
    for i in range(size_ilp * ilp, n):
        partial[0] = partial[0] + v[i]
    for k in range(1, ilp):
        partial[0] = partial[0] + partial[k]
    return partial[0].astype(dtype)


def aten_mean(x, vec, dtype, serial_only=True):
    """torch.mean of a flat tensor: sum_out(...).div_(n) in ``dtype``."""
    x = np.asarray(x)
    return (aten_sum(x, vec, dtype, serial_only) / dtype(x.shape[-1])).astype(dtype)


def aten_mean_f32(x, vec=16):
    """torch.mean over the last axis: sum_out(...).div_(n) -- one fp32 division of the fp32 sum by fp32(n)."""
    x = np.asarray(x)
    return (aten_sum_f32(x, vec) / F32(x.shape[-1])).astype(F32)
