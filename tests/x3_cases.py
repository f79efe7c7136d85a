"""Operand generators for the adversarial accuracy cases of the split-fp32 GEMM (tests/test_ops_gpu.py, tools/f32x3_adversarial.py)."""
import torch

CASES = ["elementwise_spread", "row_cancellation", "huge_times_tiny", "huge_times_huge", "tiny_times_tiny", "tiny_times_one",
         "near_fp32_max", "mixed_planes"]


def operands(case, M, N, K, seed):
    """-> (A [M,K], B [K,N]) fp32 CPU tensors.
    elementwise_spread: every ELEMENT carries its own exponent in 2^[-20, 20] (not a per-row scale).
    row_cancellation: B = [V ; -V(1 + 2^-12 u)], A = [U | U]: every result is the difference of two sums of equal size
      (|C| ~ 2^-12 |A|.|B|): whatever a plane loses shows up unmasked by larger terms.
    huge_times_tiny / huge_times_huge / tiny_times_tiny: operands near 2^+-120 / 2^+-60 whose PRODUCTS stay in fp32's range.
    tiny_times_one: operands near 2^-120 against O(1): the second and third planes fall into bf16's subnormal range.
    near_fp32_max: |a| up to 2^127 (1 - 2^-9): the first plane must not round to infinity.
    mixed_planes: values with all 24 significant bits set (x = 2^e (2 - 2^-23)), alternating signs: the worst case for
      the three 8-bit planes."""
    g = torch.Generator().manual_seed(seed)
    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    if case == "elementwise_spread":
        A = A * torch.exp2(torch.randint(-20, 21, (M, K), generator=g).float())
        B = B * torch.exp2(torch.randint(-20, 21, (K, N), generator=g).float())
    elif case == "row_cancellation":
        h = K // 2
        U, V = A[:, :h], B[:h]
        A = torch.cat([U, U], 1)
        B = torch.cat([V, -V * (1 + 2.0 ** -12 * torch.rand(h, N, generator=g))], 0)
    elif case == "huge_times_tiny":
        A, B = A * 2.0 ** 120, B * 2.0 ** -120
    elif case == "huge_times_huge":
        A, B = A * 2.0 ** 58, B * 2.0 ** 58
    elif case == "tiny_times_tiny":
        A, B = A * 2.0 ** -50, B * 2.0 ** -50
    elif case == "tiny_times_one":
        A = A * 2.0 ** -120
    elif case == "near_fp32_max":
        A = torch.sign(A) * (2.0 ** 127) * (1 - 2.0 ** -9 * torch.rand(M, K, generator=g))
        B = B * 2.0 ** -110
    elif case == "mixed_planes":
        sa = torch.where(torch.rand(M, K, generator=g) < 0.5, -1.0, 1.0)
        sb = torch.where(torch.rand(K, N, generator=g) < 0.5, -1.0, 1.0)
        A = sa * torch.exp2(torch.randint(-3, 4, (M, K), generator=g).float()) * (2 - 2.0 ** -23)
        B = sb * torch.exp2(torch.randint(-3, 4, (K, N), generator=g).float()) * (2 - 2.0 ** -23)
    else:
        raise ValueError(case)
    return A.float().contiguous(), B.float().contiguous()
