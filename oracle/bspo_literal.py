"""TEST INFRASTRUCTURE ONLY (like everything under oracle/).  Independent restatement of the reference's optimized Block Search, written statement for statement after
src/MinCostFlow.Core/Lemon/Algorithms/Internal/BlockSearchPivotOptimized.cs:23-156 (constructor, FindEnteringArc, ProcessArcRange,
ProcessArcRangeSIMD) with the two machine properties the C# reads -- Vector.IsHardwareAccelerated and Vector<long>.Count -- as
parameters.  Pure Python loops on purpose (small fixtures only): its job is to pin oracle/ns_oracle.c's opt_range_raw / block_opt_raw,
which the GPU parity tests then trust, to the control flow of the source -- in particular to what happens AFTER ProcessArcRangeSIMD
returns from a block-boundary hit (:147): the caller's scalar loop (:80) goes on with cnt == 0.

C# `int` arithmetic is 32-bit; `--cnt` running negative over at most m_s < 2^31 steps never wraps back to 0, so Python ints behave alike.
"""
import math

MIN_BLOCK_SIZE = 10          # NetworkSimplex.cs:19


class Ref:
    """`ref long min, ref int cnt, ref int bestArc`"""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v


class BlockSearchPivotOptimizedLiteral:
    def __init__(self, search_arc_num, cost, pi, source, target, state, is_hardware_accelerated, vector_long_count, block_size=0):
        # :23-36
        self.search_arc_num = search_arc_num
        block = int(math.sqrt(search_arc_num))
        self._blockSize = block_size if block_size > 0 else max(block, MIN_BLOCK_SIZE)      # block_size > 0: the tests' explicit sizes
        self._nextArc = 0
        self._costPtr, self._piPtr, self._sourcePtr, self._targetPtr, self._statePtr = cost, pi, source, target, state
        self.IsHardwareAccelerated = is_hardware_accelerated
        self.VectorLongCount = vector_long_count

    def FindEnteringArc(self):
        # :39-66
        min_ = Ref(0)
        cnt = Ref(self._blockSize)
        bestArc = Ref(-1)
        searchArcNum = self.search_arc_num
        e = self.ProcessArcRange(self._nextArc, searchArcNum, min_, cnt, bestArc)          # :49
        if e >= searchArcNum and min_.v >= 0:                                                # :52
            e = self.ProcessArcRange(0, self._nextArc, min_, cnt, bestArc)                  # :54
        if min_.v >= 0:                                                                      # :57
            return False, -1
        self._nextArc = e                                                                    # :63
        return True, bestArc.v                                                               # :64-65

    def ProcessArcRange(self, start, end, min_, cnt, bestArc):
        # :69-110
        e = start
        if self.IsHardwareAccelerated and end - start >= self.VectorLongCount * 2:           # :74
            e = self.ProcessArcRangeSIMD(start, end, min_, cnt, bestArc)                     # :76
        while e < end:                                                                       # :80
            state = self._statePtr[e]
            cost = self._costPtr[e]
            source = self._sourcePtr[e]
            target = self._targetPtr[e]
            piSource = self._piPtr[source]
            piTarget = self._piPtr[target]
            c = state * (cost + piSource - piTarget)                                         # :90
            if c < min_.v:                                                                   # :92
                min_.v = c
                bestArc.v = e
            cnt.v -= 1
            if cnt.v == 0:                                                                   # :98
                if min_.v < 0:
                    return e + 1                                                             # :102
                cnt.v = self._blockSize                                                      # :105
            e += 1
        return e                                                                             # :109

    def ProcessArcRangeSIMD(self, start, end, min_, cnt, bestArc):
        # :113-156
        vectorCount = self.VectorLongCount
        e = start
        while e <= end - vectorCount:                                                        # :119
            # :122 `costVec` is loaded and never used
            for i in range(vectorCount):                                                     # :125
                idx = e + i
                state = self._statePtr[idx]
                cost = self._costPtr[idx]
                source = self._sourcePtr[idx]
                target = self._targetPtr[idx]
                piSource = self._piPtr[source]
                piTarget = self._piPtr[target]
                c = state * (cost + piSource - piTarget)                                     # :135
                if c < min_.v:                                                               # :137
                    min_.v = c
                    bestArc.v = idx
                cnt.v -= 1
                if cnt.v == 0:                                                               # :143
                    if min_.v < 0:
                        return idx + 1                                                       # :147
                    cnt.v = self._blockSize                                                  # :150
            e += vectorCount
        return e                                                                             # :155
