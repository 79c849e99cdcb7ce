"""CSV sink for the 25-float pose messages (mirrors the surface of the reference's
``record/est_output.py``: ``EstOutputRecorder(file, tag)``, ``record_in_thread(msg_q)``, ``write_queue_to_csv(msg_q)``,
``terminate()``; one ``time`` column followed by the message slots of ``estimate/compose_msg.py:72-78``).

Host-side bookkeeping only -- no arithmetic of the path happens here.  The column names are derived from the message
layout rather than listed: slot groups (joint, what, frame) with their component letters."""
import csv
import datetime
import logging
import queue
import threading
from pathlib import Path

# (joint, kind, frame suffix) in message order: the hand rotation duplicates the lower-arm one (compose_msg.py:72)
_MSG_GROUPS = (("hand", "quat", ""), ("hand", "orig", "rh"), ("larm", "quat", "rh"), ("larm", "orig", "rh"),
               ("uarm", "quat", "rh"), ("uarm", "orig", "rh"), ("hips", "quat", "g"))
_COMPONENTS = {"quat": "wxyz", "orig": "xyz"}


def msg_columns():
    """the 25 slot names of a pose message, e.g. ``larm_quat_rh_w`` ... ``hips_quat_g_z``"""
    cols = []
    for joint, kind, frame in _MSG_GROUPS:
        stem = "_".join(p for p in (joint, kind, frame) if p)
        cols += [f"{stem}_{c}" for c in _COMPONENTS[kind]]
    return cols


class EstOutputRecorder:
    def __init__(self, file, tag: str = "REC EST OUTPUT"):
        self._path, self._tag, self._active = Path(file), tag, False
        if not self._path.parent.exists():
            raise UserWarning(f"Directory does not exist {self._path.parent}")
        with open(self._path, "w", newline="") as fd:
            csv.writer(fd).writerow(["time"] + msg_columns())
        logging.info(f"[{self._tag}] Writing to file {self._path}")

    def terminate(self):
        self._active = False

    def record_in_thread(self, msg_q: queue.Queue):
        threading.Thread(target=self.write_queue_to_csv, args=(msg_q,), daemon=True).start()

    def write_queue_to_csv(self, msg_q: queue.Queue):
        """one line per message taken from the queue: wall-clock time, then every value of the message (the reference
        writes the whole message too, est_output.py:52-57)"""
        self._active = True
        while self._active:
            try:
                msg = msg_q.get(timeout=2)
            except queue.Empty:
                logging.info(f"[{self._tag}] no data")
                continue
            row = [float(v) for v in list(msg)]
            with open(self._path, "a", newline="") as fd:
                csv.writer(fd).writerow([datetime.datetime.now()] + row)
