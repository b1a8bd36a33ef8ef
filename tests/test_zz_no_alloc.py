"""gsx_sort_pairs_u32_static -- the one entry point that allocates nothing (scratch in a device array inside the library) --
against gsx_sort_pairs_u32, through the C ABI, bit for bit (integer work), on two calls in a row, the second smaller.
Collected last (test_zz_*): the round-end run uses `pytest -x`, and a failure here must not hide the parity tests."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sort_with_workspace(lib, L, keys, values, begin_bit, end_bit):
    n = keys.numel(); dev = keys.device
    k_out = torch.empty_like(keys); v_out = torch.empty(n, dtype=torch.int32, device=dev)
    k_tmp = torch.empty_like(keys); v_tmp = torch.empty(n, dtype=torch.int32, device=dev)
    wsb = lib.gsx_sort_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    L.check(lib.gsx_sort_pairs_u32(keys.data_ptr(), None if values is None else values.data_ptr(), k_out.data_ptr(), v_out.data_ptr(),
                                   k_tmp.data_ptr(), v_tmp.data_ptr(), n, begin_bit, end_bit, 32, None, ws.data_ptr(), wsb,
                                   L.stream_ptr(dev)), "gsx_sort_pairs_u32")
    return k_out, v_out


def test_static_scratch_sort_equals_workspace_sort():
    from gaussian_splatting_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(77)
    cases = [
        # (n, begin_bit, end_bit, explicit values, key range): the largest size the scratch takes, full 32-bit keys, implicit values
        (L.STATIC_SORT_MAX_N, 0, 32, False, 1 << 31),
        # then a SMALLER one right behind it (the scratch still holds the first call's tables and buffers): ragged size,
        # 3 passes, explicit values, heavy duplicates
        (12345, 0, 24, True, 1 << 10),
        (1, 0, 32, False, 1 << 31),
        (4097, 8, 16, True, 1 << 16),                     # one pass in the middle of the key
    ]
    with L.on_device(dev):
        for n, b0, b1, with_values, hi in cases:
            keys = torch.randint(-(1 << 31) if hi == (1 << 31) else 0, hi, (n,), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
            values = torch.randperm(n, generator=g).to(torch.int32).to(dev) if with_values else None
            want_k, want_v = _sort_with_workspace(lib, L, keys, values, b0, b1)
            k_out = torch.full_like(keys, -1); v_out = torch.full((n,), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info(dev)[0]
            L.check(lib.gsx_sort_pairs_u32_static(keys.data_ptr(), None if values is None else values.data_ptr(), k_out.data_ptr(),
                                                  v_out.data_ptr(), n, b0, b1, L.stream_ptr(dev)), "gsx_sort_pairs_u32_static")
            torch.cuda.synchronize()
            if n != L.STATIC_SORT_MAX_N:                  # (the very first call on a device may load the library's code object)
                assert torch.cuda.mem_get_info(dev)[0] == free_before, "the call changed the device's free memory"
            assert torch.equal(k_out, want_k), (n, b0, b1)
            assert torch.equal(v_out, want_v), (n, b0, b1)
            # and it is a sort: keys ascend on the sorted bits, values are a permutation of the input's
            kb = (k_out.to(torch.int64) & 0xFFFFFFFF) >> b0 & ((1 << (b1 - b0)) - 1)
            assert bool((kb[1:] >= kb[:-1]).all())
            src = values if values is not None else torch.arange(n, dtype=torch.int32, device=dev)
            assert torch.equal(torch.sort(v_out.to(torch.int64)).values, torch.sort(src.to(torch.int64)).values)
        # beyond the scratch: refused with a clear status, nothing launched
        big = torch.zeros(L.STATIC_SORT_MAX_N + 1, dtype=torch.int32, device=dev)
        out_k = torch.empty_like(big); out_v = torch.empty_like(big)
        st = lib.gsx_sort_pairs_u32_static(big.data_ptr(), None, out_k.data_ptr(), out_v.data_ptr(), big.numel(), 0, 32, L.stream_ptr(dev))
        assert st == L.GSX_E_CAPACITY and b"too small" in lib.gsx_strerror(st)
        assert lib.gsx_sort_pairs_u32_static(big.data_ptr(), None, big.data_ptr(), out_v.data_ptr(), 16, 0, 32, L.stream_ptr(dev)) == L.GSX_E_INVALID
        assert lib.gsx_sort_pairs_u32_static(None, None, None, None, 0, 0, 32, L.stream_ptr(dev)) == L.GSX_OK       # empty: nothing to do
        # values sorted in place: with an odd pass count pass 0 would scatter from the buffer it scatters into -- refused
        k16 = torch.arange(16, dtype=torch.int32, device=dev); v16 = torch.arange(16, dtype=torch.int32, device=dev)
        assert lib.gsx_sort_pairs_u32_static(k16.data_ptr(), v16.data_ptr(), out_k.data_ptr(), v16.data_ptr(), 16, 8, 16, L.stream_ptr(dev)) == L.GSX_E_INVALID
        # an empty bit range copies through, and implicit values are still 0..n-1 as the header promises
        n0 = 1000
        k0 = torch.randint(0, 1 << 20, (n0,), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
        ko = torch.full_like(k0, -1); vo = torch.full((n0,), -1, dtype=torch.int32, device=dev)
        L.check(lib.gsx_sort_pairs_u32_static(k0.data_ptr(), None, ko.data_ptr(), vo.data_ptr(), n0, 12, 12, L.stream_ptr(dev)), "static, no bits")
        torch.cuda.synchronize()
        assert torch.equal(ko, k0) and torch.equal(vo, torch.arange(n0, dtype=torch.int32, device=dev))
