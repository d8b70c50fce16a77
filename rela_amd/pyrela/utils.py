"""Rate meters and the exploration schedule of the training entry point.

`generate_eps` and the `Speed:` line format are the two pieces of pyrela/utils.py the metric
depends on: env-steps/s is read off `act`, the learner rate off `train`
(pyrela/utils.py:57-74), so the line is kept parse-compatible with pyrela/parse_log.py.
"""
import time


def generate_eps(base_eps, alpha, num_actor):
    """eps_i = base_eps ** (1 + i / (N - 1) * alpha), the Ape-X schedule (pyrela/utils.py:88-96)."""
    if num_actor == 1:
        return [base_eps]
    return [base_eps ** (1 + i / (num_actor - 1) * alpha) for i in range(num_actor)]


def total_acts(actors):
    return sum(a.num_act() for a in actors)


def _short(n):
    if n < 1e3:
        return str(n)
    for div, unit in ((1e6, "M"), (1e3, "K")):
        if n >= div:
            return ("%.3f" % (n / div)).rstrip("0").rstrip(".") + unit
    return str(n)


class Tachometer:
    """Prints `Speed: train: .., act: .., buffer_add: .., buffer_size: ..` once per lap."""

    def __init__(self):
        self.seen_act = 0
        self.seen_add = 0
        self.seen_train = 0
        self.t0 = None

    def start(self):
        self.t0 = time.time()

    def lap(self, actors, replay_buffer, num_train):
        dt = time.time() - self.t0
        acts, adds = total_acts(actors), replay_buffer.num_add()
        rates = (num_train / dt, (acts - self.seen_act) / dt, (adds - self.seen_add) / dt)
        print("Speed: train: %.1f, act: %.1f, buffer_add: %.1f, buffer_size: %d" % (rates + (replay_buffer.size(),)))
        self.seen_act, self.seen_add = acts, adds
        self.seen_train += num_train
        print("Total Sample: train: %s, act: %s" % (_short(self.seen_train), _short(self.seen_act)))
        return rates
