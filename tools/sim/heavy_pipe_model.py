#!/usr/bin/env python3
"""Model of tile_heavy_pipe's step machine (softras_forward.hip): wavefront 3's decisions, the state blocks by step
parity and every double / triple buffer with an owner tag, run over random batch / round structures.  Checks that each
round is evaluated once, applied once, in order, that no buffer is written while a reader of the same step (or a later
reader of its current content) still needs it, and that the loop ends.  CPU only: python tools/sim/heavy_pipe_model.py"""
import random
import sys


def run(rounds_per_batch, verbose=False):
    nb_total = len(rounds_per_batch)             # batches the list walker will deliver
    # buffers: content tags
    rec = [None, None]; col = [None] * 4; M = [None, None]
    cell = [None, None]; pair = [None, None]; span = [None] * 3
    state = [dict(valid=0, done=0, masks=0), dict(valid=0, done=0, masks=0)]
    log_eval, log_apply = [], []
    # prologue
    staged_batches = 0

    def stage(nb):
        nonlocal staged_batches
        if staged_batches >= nb_total:
            return 0
        assert nb == staged_batches, (nb, staged_batches)
        staged_batches += 1
        rec[nb & 1] = nb; col[nb & 3] = nb
        return 1
    if stage(0) == 0:
        return log_eval, log_apply
    M[0] = 0

    def build_list(nb, r, ns, st):
        assert M[nb & 1] == nb, ("list reads masks of another batch", nb, M)
        last = r + 1 >= rounds_per_batch[nb]
        st.update(valid=1, batch=nb, r=r, last=last)
        span[ns % 3] = (nb, r); pair[ns & 1] = (nb, r)
    build_list(0, 0, 0, state[0])
    state[0].update(masks=0, done=0)
    cur_batch, staged, masked, walker_done, offer_next, offer_now = 0, False, False, False, False, False
    a = None
    for step in range(10000):
        st, nx = state[step & 1], state[(step + 1) & 1]
        e = (st["batch"], st["r"]) if st["valid"] else None
        if st["done"] and a is None:
            break
        reads = []                      # (buffer name, index, expected tag) read during this step
        writes = []                     # (buffer name, index)
        if a is not None:
            reads += [("span", (step - 1) % 3, a), ("cell", (step - 1) & 1, a), ("col", a[0] & 3, a[0])]
            log_apply.append(a)
        # wavefront 3
        if offer_now:
            masked = True
        offer_now, offer_next = offer_next, False
        nmasks = 0
        if not staged and not walker_done and not st["done"]:
            nbn = cur_batch + 1
            writes += [("rec", nbn & 1), ("col", nbn & 3)]
            if stage(nbn) == 0:
                walker_done = True
                writes = writes[:-2]
            else:
                staged, nmasks, offer_next = True, 1, True
        listed = False
        nxnew = dict(nx)
        if e is not None and not st["last"]:
            writes += [("span", (step + 1) % 3), ("pair", (step + 1) & 1)]
            build_list(e[0], e[1] + 1, step + 1, nxnew); listed = True
        elif masked:
            cur_batch += 1; staged = masked = False
            writes += [("span", (step + 1) % 3), ("pair", (step + 1) & 1)]
            build_list(cur_batch, 0, step + 1, nxnew); listed = True
        if not listed:
            nxnew["valid"] = 0
        nxnew.update(masks=nmasks, mbatch=cur_batch + 1, done=int(not listed and walker_done and not staged))
        # tasks
        if e is not None:
            reads += [("pair", step & 1, e), ("rec", e[0] & 1, e[0])]
            writes += [("cell", step & 1)]
            log_eval.append(e)
        if st["masks"]:
            mb = st["mbatch"]
            reads += [("rec", mb & 1, mb)]
            writes += [("M", mb & 1)]
        # hazards: a buffer written in this step must not be read in this step
        bufs = dict(rec=rec, col=col, M=M, cell=cell, pair=pair, span=span)
        for name, idx in writes:
            for rn, ri, tag in reads:
                assert not (rn == name and ri == idx), ("step %d: %s[%d] written while read" % (step, name, idx))
        # reads see the expected content (checked against the state BEFORE this step's writes, except what was just staged)
        for rn, ri, tag in reads:
            got = bufs[rn][ri]
            if rn in ("span", "pair", "cell"):
                assert got == tag, ("step %d: %s[%d] holds %s, reader expects %s" % (step, rn, ri, got, tag))
            else:
                assert got == tag, ("step %d: %s[%d] holds batch %s, reader expects %s" % (step, rn, ri, got, tag))
        if e is not None:
            cell[step & 1] = e
        if st["masks"]:
            M[st["mbatch"] & 1] = st["mbatch"]
        state[(step + 1) & 1] = nxnew
        a = e
        if verbose:
            print(step, "E", e, "A", log_apply[-1:] , "staged", staged, "masked", masked, "done", nxnew["done"])
    else:
        raise AssertionError("no termination")
    return log_eval, log_apply


if __name__ == "__main__":
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    for trial in range(20000):
        nb = rng.randint(1, 7)
        rp = [rng.choice([1, 1, 2, 3, 4, 6]) for _ in range(nb)]
        ev, ap = run(rp)
        want = [(b, r) for b in range(nb) for r in range(rp[b])]
        assert ev == want, (rp, ev)
        assert ap == want, (rp, ap)
    print("ok: 20000 random batch / round structures")
