# assembles profiles/r06_sampler_quality.txt from the outputs of tools/r06_quality_sweep*.sh and tools/r06_block_speed*.sh (gpurun_out/r06)
O=gpurun_out/r06
{
cat <<'E'
# Round 6: what the ORDER of the fast samplers costs in ranking quality (tools/sampler_quality.py; tools/r06_quality_sweep*.sh, tools/r06_block_speed*.sh).
# The round-5 review: the headline's speed "comes from the sampler's stratification (negatives of ~20 consecutive positions drawn from one 2-item block), a
# documented divergence from i.i.d. negatives whose only evidence of harmlessness is two planted-factor ranking tests".  This is the evidence, for and against.
#
# Planted-factor dataset in the headline's proportions (one positive per user per step, batch = users = 20 x items, popular head; 15 train + 5 held-out
# positives per user), the model class's own fit + Evaluator, same number of steps, several seeds per arm; NDCG@10 / Recall@10 on the held-out positives.
#   iid = independent uniform negatives (the reference's sampler, data/generators.py:151-201, on the device) . blocked = the headline's layout (item block c = 2 in
#   these runs) . ranges = two item ranges (bench.py's N > 1 default, c = 3) . csc = the blocked layout with the round-6 walk drawing the positives
#
# FINDING: the stratified layouts train to the same quality at a SMALL step size and fall short by 1-2.5 % of NDCG@10 at a LARGE one.  Same expectation as
# independent draws (every user's negative is uniform), more variance per step: the negatives of one block all come from the ~c B / I consecutive positions
# of one wavefront, i.e. from users who share a positive item -- a coherent push on the block's rows instead of ~20 unrelated ones.  It shrinks with the
# block (2.4 / 1.4 / 1.4 / 0.9 / 0.8 / 0.45 % at c = 2 / 3 / 4 / 6 / 8 / 16) and with the step size (2.4 % at 0.1 per triplet, 0.7 % at 0.05, 0.2 % = inside the seed
# noise at 0.02 -- and 0.02 is where this model is BEST: 0.1245 against 0.1105 at 0.1).  The fast arms (blocked, ranges, csc) are alike.  d = 128 (overfits after
# 300 steps): 0.7 %.  400 000 x 20 000: 2.0 % at 0.1.
# Speed side, one box, alternating (us per step at the headline): c = 2: 307 / 329 / 321 . c = 3: 303 / 322 / 322 . c = 4: 327 / 336 / 337 . c = 5: 341 / 343 / 356
# (single pass on an earlier box: 306 / 307 / 308 / - / 355 (c = 6) / 359 (c = 8) / 342 (12) / 374 (16)); d = 64: c = 2 / 3 / 4 alike; two item ranges: c = 3 / 4 / 5 alike.
# CONSEQUENCES: (1) the floor of the block moved from 2 to 3 (sharded.pick_neg_block: free on the clock, closes 40 % of the gap; the profiles of the round
# were re-taken with it); (2) hparams['neg_block_min'] / BPREngine.set_neg_block(min_block=) give larger blocks, ~4 % of the step per size from 4 on;
# (3) hparams['neg_block'] = 0 is the reference's independent sampler (1.6e9 triplets/s at the headline shape instead of 3.0-3.3e9).
#
E
echo "## 16 seeds, 600 steps, step size 0.1 per triplet, all four arms"; grep "^#" $O/sampler_quality_600.txt
echo "## 6 seeds, 150 steps (the first run)"; grep "^#" $O/sampler_quality.txt
for n in long lr05 lr02 lr02_16 d128 big; do echo "## sweep: $n"; grep "^#" $O/sq_$n.txt; done
for c in 2 3 4 6 8 16; do echo "## item block forced to $c (1000 steps, 8 seeds, 0.1 per triplet)"; grep "^#" $O/sq_c$c.txt | tail -4; done
echo "## independent negatives, the same 1000 steps"; grep "^#" $O/sq_iid1000.txt | tail -4
echo "## AT THE HEADLINE SHAPE (tools/r06_quality_headline.sh: 1M users x 100K items, d = 128, batch = users, item blocks of 3, 0.05 per triplet, 3 seeds), with the clock:"
echo "## same NDCG@10 step for step; 400 steps take 0.24 s of training blocked, 0.44 s with independent negatives -- the same quality 1.86x sooner"
grep "^#" $O/sq_headline_lr05.txt
echo "## headline step time against the item block, first box (RSX_NEG_BLOCK_EXACT, bench.py --no-legs)"; cat $O/block_speed.txt
echo "## ... alternating on a second box: headline, d = 64, two item ranges"; cat $O/block_speed2.txt
} > profiles/r06_sampler_quality.txt
