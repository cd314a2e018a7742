"""Label re-timing against vectors produced by the reference (oracle/gen_golden.py: gen_labels)."""
import os

import numpy as np
import pytest

from magphase_amd import libaudio as la
from magphase_amd import magphase as mp

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_labels.npz"))


@pytest.mark.parametrize("case", ["exact", "short_end", "too_short"])
@pytest.mark.parametrize("pz", [0, 1])
def test_frames_per_state_and_relabel(tmp_path, case, pz):
    lab = tmp_path / (case + ".lab")
    lab.write_text(str(G["lab_" + case]))
    key = "%s_pz%d" % (case, pz)
    if "error_" + key in G.files:
        with pytest.raises(ValueError) as e:
            mp.get_num_of_frms_per_state(G["v_shift"], str(lab), int(G["fs"]), b_prevent_zeros=bool(pz))
        assert str(e.value) == str(G["error_" + key])
        return
    v_n = mp.get_num_of_frms_per_state(G["v_shift"], str(lab), int(G["fs"]), b_prevent_zeros=bool(pz))
    assert v_n.dtype == np.float64 and np.array_equal(v_n, G["nfrms_" + key])
    out = tmp_path / "out.lab"
    la.convert_label_state_align_to_var_frame_rate(str(lab), v_n, str(out))
    assert out.read_text() == str(G["outlab_" + key])


def test_phone_without_frames_raises(tmp_path):
    # 2 phones x 5 states; the second phone lies after the last epoch but within the tolerance rule's reach
    lines, t = [], 0
    for i in range(10):
        dur = 500000 if i < 5 else 50000
        lines.append("%d %d p[%d]" % (t, t + dur, i % 5 + 2))
        t += dur
    lab = tmp_path / "x.lab"
    lab.write_text("\n".join(lines) + "\n")
    v_shift = np.full(40, 240)          # 40 epochs, 5 ms apart: all inside the first phone (250 ms)
    with pytest.raises(ValueError, match="do\\(es\\) not contain any frame"):
        mp.get_num_of_frms_per_state(v_shift, str(lab), 48000)
