"""CPU tier: limo_ba_evaluate_rows of the emulated pipeline (the lane functions gp_lane / reg_row_eval compiled for the host +
the row assembly of kba_rows.hpp) against the oracle's dual-number rows.  The GPU tier runs the same cases through
liblimo_hip.so (tests/test_gpu_ba.py::test_ground_and_regulariser_rows_match_oracle)."""
import rows_common


def test_ground_and_regulariser_rows_match_oracle(emu, oracle):
    rows_common.run(emu.evaluate_rows, oracle)
