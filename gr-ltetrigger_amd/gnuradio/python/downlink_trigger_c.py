"""downlink_trigger_c on GNU Radio's own runtime: one 1.92 Msps complex input, message ports "track" and "drop".

Interface of the reference's hier block (python/downlink_trigger_c.py:13-73): constructor (psr_threshold,
exit_on_success), set_psr_threshold with the 1.5 floor, members pss0..2 / sss0..2 / mib0..2.  NOT RUN IN THIS
REPOSITORY (no GNU Radio); the same topology on the in-repo runtime is gr-ltetrigger_amd/downlink_trigger_c.py, which
the test suite exercises.
"""
from gnuradio import blocks, gr

from . import blocks_module as _blocks  # pybind11 or SWIG, resolved once in __init__.py

MIN_PSR_THRESHOLD = 1.5  # the reference clamps here: lower thresholds crashed its srsLTE calls


class downlink_trigger_c(gr.hier_block2):
    def __init__(self, psr_threshold, exit_on_success=False):
        gr.hier_block2.__init__(self, "downlink_trigger_c",
                                gr.io_signature(1, 1, gr.sizeof_gr_complex), gr.io_signature(0, 0, 0))
        self.psr_threshold = max(psr_threshold, MIN_PSR_THRESHOLD)
        self.exit_on_success = exit_on_success
        for port in ("drop", "track"):  # both directions, as the reference does (python/downlink_trigger_c.py:47-48,55-56)
            self.message_port_register_hier_in(port)
            self.message_port_register_hier_out(port)
        self._chains = []
        for n_id_2 in range(3):
            pss = _blocks.pss(n_id_2, self.psr_threshold)
            sss = _blocks.sss(n_id_2)
            mib = _blocks.mib(exit_on_success)
            sink = blocks.tag_debug(gr.sizeof_gr_complex, "mib%d" % n_id_2)
            sink.set_display(False)
            self.connect(self, pss, sss, mib, sink)
            for port in ("track", "drop"):
                self.msg_connect(mib, port, self, port)
            setattr(self, "pss%d" % n_id_2, pss)
            setattr(self, "sss%d" % n_id_2, sss)
            setattr(self, "mib%d" % n_id_2, mib)
            self._chains.append((pss, sss, mib, sink))

    def set_psr_threshold(self, t):
        self.psr_threshold = max(t, MIN_PSR_THRESHOLD)
        for pss, _, _, _ in self._chains:
            pss.set_psr_threshold(self.psr_threshold)
