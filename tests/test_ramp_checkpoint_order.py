"""A redone (-B) job whose checkpoint list ends up out of order only in its wrongly aligned tail: the reference's checks at the end of
getSqrtSlices (GraphAligner.h:2833-2842) see the whole list -- removeWronglyAlignedEnd trims it afterwards (:3002,3018) -- so the read
fails its assertion.  Round 1's device code trimmed first and reported such reads as aligned; a randomised campaign on the MI355X
(seed 889, trial 66, read 7; 4 of 28 800 reads over two campaigns) found it in round 2.  This is that trial on the host emulation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_campaign_trial_with_out_of_order_checkpoints_in_the_trimmed_tail():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import parity_campaign
    stats = parity_campaign.main(["--emul", "--trials", "67", "--reads", "24", "--seed", "889", "--only", "66"], quiet=True)
    assert stats["ramp_trials"] >= 1 and stats["compared"] == 24
    assert stats["mismatches"] == 0, stats["first_mismatches"]
    assert stats["dev_status"].get("1", 0) >= 1          # the read in question reports the reference's assertion
