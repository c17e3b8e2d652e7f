#!/usr/bin/env python3
"""Search for LTE cells in a recorded IQ file (complex float32) -- counterpart of the reference's
examples/cell_search_file.py:33-138 on the MI355X path.

Example:
  $ ./cell_search_file.py -s 15.36M --repeat -c 15.36M tests/golden/test_frames/lte_frame_50prb_cellid_125
  Starting cell search... done.
  {
      "cell_id": 125,
      "nof_tx_ports": 1,
      "cp_len": "Normal",
      "nof_prb": 50,
      "phich_len": "Normal",
      "nof_phich_resources": "1",
      "sfn_offset": 0,
      "tracking_start_time": 1471540412,
      "status": "FOUND"
  }
"""
import argparse
import json
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import ltetrigger_amd as ltetrigger  # noqa: E402

REQUIRED_SAMPLE_RATE = 1.92e6

_ENG = {"k": 1e3, "M": 1e6, "G": 1e9, "m": 1e-3, "u": 1e-6}


def str_to_num(value):
    """Engineering notation ('15.36M') -> float, like gnuradio.eng_notation.str_to_num."""
    value = value.strip()
    if value and value[-1] in _ENG:
        return float(value[:-1]) * _ENG[value[-1]]
    return float(value)


class cell_search_file(ltetrigger.TopBlock):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.logger = logging.getLogger("cell_search_file")
        if args.sample_rate % REQUIRED_SAMPLE_RATE:
            self.logger.error("Sample rate %.2f MHz is not a multiple of 1.92 MHz. "
                              "Arbitrary resampling not supported at this time.", args.sample_rate / 1e6)
            sys.exit(-1)
        resamp_ratio = int(args.sample_rate / REQUIRED_SAMPLE_RATE)

        lastblock = self.file_source(args.filename, repeat=args.repeat)
        if resamp_ratio != 1:
            self.resampler = self.rational_resampler_ccc(1, resamp_ratio)
            self.connect(lastblock, self.resampler)
            lastblock = self.resampler
        # --throttle only lowers CPU load in the reference; a file is processed as fast as possible here
        if args.cut_off > -1:
            cut_off = self.head(args.cut_off)
            self.connect(lastblock, cut_off)
            lastblock = cut_off
        self.trigger = ltetrigger.downlink_trigger_c(self, psr_threshold=args.threshold, exit_on_success=True)
        self.cellstore = self.cellstore()
        self.connect(lastblock, self.trigger)
        self.msg_connect(self.trigger, "track", self.cellstore, "track")
        self.msg_connect(self.trigger, "drop", self.cellstore, "drop")


def main(args):
    logging.basicConfig()
    tb = cell_search_file(args)
    print("Starting cell search... ", end="")
    sys.stdout.flush()
    tb.run(args.time_out if (args.cut_off == -1 and args.time_out > -1) else -1.0)
    print("done.")

    results = []
    if tb.cellstore.tracking():
        for cell in tb.cellstore.cells():
            cell["status"] = "FOUND"
            results.append(json.dumps(cell, indent=4))
    else:
        results.append(json.dumps({"status": "NOT_FOUND"}))
    for cell in results:
        print(cell)

    if args.fifoname:
        if not os.path.exists(args.fifoname):
            os.mkfifo(args.fifoname)
        pipeout = os.open(args.fifoname, os.O_WRONLY)
        for cell in results:
            os.write(pipeout, (str(len(cell)) + "\n" + cell).encode())
        os.close(pipeout)
    return results


def parse_args(argv=None):
    def eng_float(v):
        try:
            return str_to_num(v)
        except ValueError:
            raise argparse.ArgumentTypeError(f"invalid engineering notation value: {v!r}")

    def eng_int(v):
        return int(eng_float(v))

    def filetype(fname):
        if os.path.isfile(fname):
            return fname
        raise argparse.ArgumentTypeError(f"file {fname} does not exist")

    p = argparse.ArgumentParser()
    p.add_argument("filename", type=filetype)
    p.add_argument("-s", "--sample-rate", type=eng_float, required=True, metavar="Hz",
                   help="input data's sample rate [Required]")
    p.add_argument("-f", "--frequency", type=eng_float, metavar="Hz", help="input data's center frequency")
    p.add_argument("--repeat", action="store_true",
                   help="loop file until cell found or cut-off reached [default=%(default)s]")
    p.add_argument("-c", "--cut-off", type=eng_int, metavar="N", default=-1,
                   help="stop looping after N samples [default=%(default)s]")
    p.add_argument("--throttle", type=eng_float, metavar="Hz", help="accepted for compatibility; ignored")
    p.add_argument("--time-out", type=eng_float, metavar="sec", default=-1,
                   help="max time in seconds to perform search [default=%(default)s]")
    p.add_argument("--threshold", type=eng_float, default=4,
                   help="set peak to side-lobe ratio threshold [default=%(default)s]")
    p.add_argument("--fifoname", default=None, help="FIFO name to which to write output")
    return p.parse_args(argv)


if __name__ == "__main__":
    main(parse_args())
