"""downlink_trigger_c -- hierarchical block for LTE downlink detection, input at 1.92 Msps.

Python counterpart of the reference's python/downlink_trigger_c.py:13-73 on the in-repo runtime:
three chains pss(N_id_2=k) -> sss(k) -> mib -> sink, message ports "track" and "drop" fanned in
from the three mib blocks, set_psr_threshold fanned out with the same 1.5 floor.
"""
MIN_PSR_THRESHOLD = 1.5  # below this the reference reports crashes (python/downlink_trigger_c.py:10)


class downlink_trigger_c:
    def __init__(self, tb, psr_threshold, exit_on_success=False):
        self.tb = tb
        self.psr_threshold = self._ensure_safe_threshold(psr_threshold)
        self.exit_on_success = exit_on_success
        self.pss = [tb.pss(N_id_2=k, psr_threshold=self.psr_threshold) for k in range(3)]
        self.sss = [tb.sss(N_id_2=k) for k in range(3)]
        self.mib = [tb.mib(exit_on_success=exit_on_success) for _ in range(3)]
        self.tag = [tb.null_sink() for _ in range(3)]  # tag_debug with display off
        for k in range(3):
            tb.connect(self.pss[k], self.sss[k], self.mib[k], self.tag[k])
        self.pss0, self.pss1, self.pss2 = self.pss

    # hier-block plumbing used by TopBlock.connect / msg_connect
    def _hier_connect_input(self, tb, upstream):
        for k in range(3):
            tb.connect(upstream, self.pss[k])

    def _hier_msg_sources(self, port):
        if port not in ("track", "drop"):
            raise ValueError(f"no message port {port!r}")
        return self.mib

    def set_psr_threshold(self, t):
        t = self._ensure_safe_threshold(t)
        self.psr_threshold = t
        for p in self.pss:
            p.set_psr_threshold(t)

    @staticmethod
    def _ensure_safe_threshold(t):
        return t if t > MIN_PSR_THRESHOLD else MIN_PSR_THRESHOLD
