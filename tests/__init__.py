"""Test suite of captra_amd: CPU tests (oracle, goldens, host logic, C-ABI symbols) and @pytest.mark.gpu parity tests."""
