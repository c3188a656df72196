"""API surface of the reference's utils/perfect_repeat_tracker.py, kept so that code importing
`PerfectRepeatTracker` / `consists_of_perfect_repeats` keeps working.

This is NOT the scan path of this build: `detect_repeats()` runs on the GPU through libprf and never
instantiates this class.  The class is a small step-at-a-time model with the same observable behaviour as
the reference's (one base per advance() call, rows written into a shared dict keyed by (start, end)), for
callers that drive a tracker by hand on short strings.  Written from the behaviour described in
SURVEY.md section 3.3, not from the reference's text.
"""


def consists_of_perfect_repeats(sequence):
    """Smallest unit u (len(u) < len(sequence)) with sequence == u * n, or None if the word is primitive.
    Same contract as reference utils/perfect_repeat_tracker.py:108-142."""
    n = len(sequence)
    if n < 2:
        return None
    # a word is a power of a shorter word iff it occurs inside its own doubling at an offset 0 < p < n
    p = (sequence + sequence).find(sequence, 1)
    return sequence[:p] if p < n else None


class PerfectRepeatTracker:
    """Follows one motif size through a sequence, one position per advance()."""

    def __init__(self, motif_size, min_repeats, min_span, input_sequence, output_intervals, verbose=False):
        self.motif_size = motif_size
        self.min_repeats = min_repeats
        self.min_span = min_span
        self.input_sequence = input_sequence
        self.output_intervals = output_intervals
        self.verbose = verbose
        self._current_position = 0
        self._run_length = 1      # 1 + number of consecutive matches ending just before the current position

    @property
    def current_position(self):
        return self._current_position

    def log(self, message, force=False):
        if not (force or self.verbose):
            return
        begin = max(0, self._current_position - self._run_length)
        motif = self.input_sequence[begin:begin + self.motif_size]
        shown = self.input_sequence[begin:self._current_position + 1]
        if self._current_position - begin >= 300:
            shown = "[too long]"
        copies = (self._current_position - begin) / len(motif)
        print(f"{message:100s}  || PerfectRepeatTracker:{len(self.input_sequence):,d}bp  "
              f"[{begin}:{self._current_position + 1}], run={self._run_length}, i0={self._current_position}: "
              f"{copies:0.2f} x {motif} ==> {shown}")

    def _matches_here(self):
        s, i, k = self.input_sequence, self._current_position, self.motif_size
        return s[i] == s[i + k] and s[i] != "N"

    def advance(self):
        """One step to the right; False once the end of the comparable range is reached."""
        if self._current_position >= len(self.input_sequence) - self.motif_size:
            return False
        if self._matches_here():
            self._run_length += 1
        else:
            self.output_interval_if_it_passes_filters()
            self._run_length = 1
        self._current_position += 1
        return True

    def is_in_middle_of_repeat(self):
        return self._run_length > self.motif_size

    def done(self):
        self.output_interval_if_it_passes_filters()

    def _long_enough(self, covered):
        return covered >= self.min_span and covered >= self.min_repeats * self.motif_size

    def output_interval_if_it_passes_filters(self):
        """Called where a run of matches ends: record it if it is long enough and its motif is primitive."""
        s, k = self.input_sequence, self.motif_size
        last = self._current_position
        first = last - self._run_length + 1
        motif = s[first:first + k]
        if "N" in motif:
            return
        if self._long_enough(self._run_length + k - 1):
            # the k-1 bases after the last compared position belong to the final copy of the motif
            while last < len(s) - 1 and s[last + 1] == s[last + 1 - k]:
                last += 1
                self._run_length += 1
        if not self._long_enough(self._run_length):
            return
        key = (first, last + 1)
        known = self.output_intervals.get(key)
        if known is not None and len(known) < len(motif):
            return
        if consists_of_perfect_repeats(motif) is None:
            self.output_intervals[key] = motif
