#!/usr/bin/env python3
"""Condense the output of tools/ubench/aql_chain (device-kernarg run + host-kernarg run) into profiles/r04_chain_aql.txt.

usage: python tools/aql_chain_summary.py gpurun_out/aql_chain_dev.txt gpurun_out/aql_chain_host.txt > profiles/r04_chain_aql.txt
"""
import re
import sys

HEADER = '''r04 -- can DEPENDENT launches overlap on gfx950?  (VERDICT r3 "Next" item 1, steps a and b: measured -- the hardware serialises)
tool: tools/ubench/aql_chain.hip (own HSA user queues, code object through the HSA loader, hand-written AQL packets, one doorbell per
batch, s_memrealtime stamps + XCC_ID per workgroup); 1 x MI355X, ROCm 7.2; `./aql_chain aql_chain.hsaco dev` (kernargs of the own
queue in device memory; the host-coherent run is at the end: there every stage pays ~1.7 us more -- a cold kernarg fetch over PCIe
per packet -- which is why HIP keeps kernargs in device memory on this part).

FINDINGS
 1. ONE queue never overlaps two dispatches, barrier bit or not.  With barrier bit 0, fence scope NONE, 256 workgroups of 512
    threads per stage (one of the four slots of every CU: three more stages WOULD fit) and a 4 us body, stage k+1's first workgroup
    on an XCD starts +0.64 .. +0.88 us AFTER stage k's last workgroup on THAT XCD has ended, for all 159 x 8 (stage, XCD) pairs
    ("SAME XCD min" column of "own queue, barrier 0, independent"); a stage costs 5.07 us against 5.44 under a hipGraph.  What looks
    like overlap chip-wide ("chip-wide p50" negative) is the eight XCDs drifting apart: each XCD walks the packet stream on its own
    (empty stages: 20 packets apart).  hip_ext.h:67 ("hipExtAnyOrderLaunch is not supported on GFX9xx") describes the packet
    processor, not HIP: hipExtLaunchKernelGGL with that flag behaves like the own queue with barrier 0 (section a).
 2. The launch-order property a flag-ordered chain would need holds PER XCD (section b, 10 / 10: no B workgroup enters an XCD
    before all of A's workgroups of that XCD have entered; chip-wide it does not hold, 0 / 10) -- and is moot by finding 1.
 3. The queue barrier is as cheap as any chip-wide hand-off.  A chain of EMPTY 768-workgroup stages costs 1.80 us per stage under
    a hipGraph (1.45 on the own queue with fence scope NONE, 1.87 with AGENT); the same chain ordered by in-memory epoch counters
    instead (32 arrival groups + top + 32 done words, one polling lane per workgroup: the engine's protocol) costs 2.46 on one
    queue and 1.65 on TWO queues, where waiting successors are resident beside their predecessor.  (A flat counter: 768 x 13 ns =
    10 us per stage, first version of the tool.)
 4. DIFFERENT queues do overlap (two queues, independent 256-workgroup stages of 4 us: 2.53 us per stage; four queues: 1.29).
    A flag-ordered chain alternating between two queues beats the hipGraph only when TWO whole stages are resident together:
    256 / 384-workgroup stages, 1 us prologue + 3 us body: 4.70 / 4.75 vs 5.44 / 5.48 us per stage -- the pre-wait prologue hides,
    the hand-off itself (1.65 us) is no cheaper than the queue barrier's 1.1 us + 0.4 us of launch skew.  The decode GEMV occupies
    the chip (768 workgroups x 8 waves x 80 VGPRs = 6 waves per SIMD): a successor workgroup can only enter when a predecessor's
    retires (slot turnaround 0.9 - 1.2 us), what it could do before its wait (kernarg, weight priming: ~0.3 us of issue) is small
    against the hand-off, and two queues give NO launch order, so a waiting successor can strand its predecessor's unlaunched
    workgroups unless every stage is cut to half the chip -- the persistent engine's occupancy loss again (DESIGN 3.2b: 510 vs 320
    SIMD-cycles per tile at 4 waves per SIMD).
 5. What an own queue would buy: fence scope NONE saves 0.14 us per stage against the hipGraph (5.53 vs 5.67 at 768 workgroups,
    1 + 3 us) = 23 us per token = 1.9 %, at the price of sc1 hand-offs in every kernel of the step; host cost of one submission of
    160 packets 9 - 10 us against 17 us for hipGraphLaunch (hidden either way: the host runs ahead of a 1.2 ms token).
 => the five-launch hipGraph step stays.  A dependent stage on this chip costs >= 1.45 us of hand-off whatever performs it; the other
    ~3 us per launch sit inside the kernel (x staging, first data, tail) and can only be hidden by co-residency that the GEMV's
    occupancy does not leave room for.

SUMMARY (us per stage; 160 stages per submission; prologue P = work before the wait, body T = work after it; kernargs in device memory)
'''


def summarize(txt):
    lines = txt.split('\n')
    out = []
    for i, l in enumerate(lines):
        if l.startswith('=== ') or l.startswith(' prologue'):
            out.append(l)
        m = re.match(r'\s+(.{46})\s+(\d+) stages: device span\s+[\d.]+ us =\s+([\d.]+) us/stage .*first-start - prev-last-end p50\s+'
                     r'([+-][\d.]+).*host wall\s+([\d.]+)', l)
        if m:
            m2 = re.search(r'start skew\s+([\d.]+) \| LAST start - prev last end ([+-][\d.]+).*SAME XCD, first start - prev last end: '
                           r'min ([+-][\d.]+) p50 ([+-][\d.]+) max ([+-][\d.]+)', lines[i + 1])
            out.append(f"   {m.group(1).strip():46s} {m.group(3):>7s} us/stage | start skew p50 {m2.group(1):>5s} | first start - prev last end: "
                       f"chip-wide p50 {m.group(4):>7s}, LAST start {m2.group(2):>6s} | SAME XCD min {m2.group(3)} p50 {m2.group(4)} max {m2.group(5)} "
                       f"| host wall {m.group(5):>7s} us")
    return '\n'.join(out)


def main():
    dev = open(sys.argv[1]).read()
    host = open(sys.argv[2]).read()
    mark = '=== 768 workgroups per stage'
    print(HEADER + summarize(dev))
    print('\nRAW OUTPUT, sections (a) and (b), device-kernarg run\n' + dev.split(mark)[0])
    print('SUMMARY, kernargs of the own queue in HOST-COHERENT memory (768-workgroup stages only; the hipGraph rows are HIP\'s own kernargs)')
    print(summarize(mark + host.split(mark)[1].split('=== 256 workgroups')[0]))


if __name__ == '__main__':
    main()
