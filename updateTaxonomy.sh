#!/bin/sh
# updateTaxonomy.sh — refreshes the taxonomy dumps of the database directory recorded in ./.DBDirectory (needs network
# access; CuCLARK's updateTaxonomy.sh:24-58).  The per-database metadata (<dir>/.<db>.fileTo*) is rebuilt by the next
# set_targets.sh run once those files are removed.
HERE=$(dirname "$0")
if [ ! -s ./.DBDirectory ]; then
  echo "There is no database directory: run set_targets.sh first."
  exit 1
fi
while read -r DIR; do
  [ -n "$DIR" ] && "$HERE/download_taxondata.sh" "$DIR/taxonomy"
done < ./.DBDirectory
