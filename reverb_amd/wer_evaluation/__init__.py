"""WER evaluation glue of the reference (asr/wer_evaluation/): `scoring_commands` writes one alignment command per
hypothesis CTM, `aggregate_scoring` sums the per-file JSON logs into WER / insertion / deletion / substitution rates.
The reference delegates the alignment itself to the external `fstalign` binary; `align` is a built-in stand-in for
plain word sequences (native Levenshtein counts in librvb, no synonym / normalisation FSTs) that writes the same
`wer.bestWER` JSON block, so the two scripts work end to end without it."""
