#!/bin/sh
# make_metadata.sh — per-database metadata for set_targets.sh; leaves the same files as CuCLARK's script of this name
# (make_metadata.sh:24-125), so database directories prepared by either can be used by both:
#   ./make_metadata.sh <bacteria|viruses|human|custom> <database directory>
# <dir>/.<db>                     list of sequence files (custom: every <dir>/Custom/*.f*)
# <dir>/.<db>.fileToAccssnTaxID   file, accession, taxonomy ID           (exe/getAccssnTaxID)
# <dir>/.<db>.fileToTaxIDs        file, taxonomy ID, species..phylum IDs  (exe/getfilesToTaxNodes)
# Taxonomy dumps live in <dir>/taxonomy (nodes.dmp, merged.dmp, nucl_accss), marked complete by <dir>/.taxondata.
# Missing dumps or sequences are fetched by download_taxondata.sh / download_data.sh (network access needed).
HERE=$(dirname "$0")

stop() { echo "$1"; exit 1; }

[ $# -ge 2 ] && [ -n "$2" ] || { echo "Usage: $0 <Database: bacteria, viruses, human or custom> <Directory path>"; exit 0; }
DB=$1
DBDR=$2
TAX="$DBDR/taxonomy"
LIST="$DBDR/.$DB"
ACC="$DBDR/.$DB.fileToAccssnTaxID"
LINEAGE="$DBDR/.$DB.fileToTaxIDs"

# -- directories
if [ ! -d "$DBDR" ]; then
  echo "Database directory $DBDR does not exist yet: creating it."
  mkdir -m 775 "$DBDR" || stop "Cannot create $DBDR (check the path)."
fi
[ -d "$DBDR/Custom" ] || mkdir -m 775 "$DBDR/Custom"

# -- taxonomy dumps
if [ ! -f "$DBDR/.taxondata" ]; then
  echo "No taxonomy dumps in $TAX yet: downloading them."
  mkdir -p -m 775 "$TAX"
  "$HERE/download_taxondata.sh" "$TAX"
  [ -f "$DBDR/.taxondata" ] || stop "Taxonomy dumps are still missing (nodes.dmp, merged.dmp, nucl_accss in $TAX). Giving up."
fi

# -- list of sequence files
if [ ! -s "$LIST" ]; then
  case "$DB" in
    custom)
      find "$DBDR/Custom/" -name '*.f*' > "$LIST"
      if [ ! -s "$LIST" ]; then
        echo "The database directory 'Custom' is empty."
        echo "To classify against your own sequences:"
        echo "  1) put them (FASTA, header starting with the accession number) into $DBDR/Custom/"
        echo "  2) run this command again with the database 'custom'"
        exit 1
      fi ;;
    *)
      echo "No $DB sequences in $DBDR yet: downloading them."
      "$HERE/download_data.sh" "$DBDR" "$DB" ;;
  esac
fi
[ -s "$LIST" ] || stop "There are no $DB sequences to work with. Giving up."
for tool in getAccssnTaxID getfilesToTaxNodes; do
  [ -x "$HERE/exe/$tool" ] || stop "exe/$tool is missing: build the tools first (make -C cuclark_amd/csrc)."
done

# -- file -> accession -> taxonomy ID -> lineage
if [ "$DB" = "human" ]; then   # one fixed lineage: Homo sapiens, Homo, Hominidae, Primates, Mammalia, Chordata
  if [ ! -s "$LINEAGE" ]; then
    while read -r file; do echo "$file X 9606 9605 9604 9443 40674 7711"; done < "$LIST" > "$LINEAGE"
  fi
  exit 0
fi
if [ ! -s "$ACC" ]; then
  echo "$DB: looking up the accession and taxonomy ID of every file..."
  "$HERE/exe/getAccssnTaxID" "$LIST" "$TAX/nucl_accss" "$TAX/merged.dmp" > "$ACC"
fi
if [ ! -s "$LINEAGE" ]; then
  echo "$DB: walking the taxonomy tree for every file..."
  "$HERE/exe/getfilesToTaxNodes" "$TAX/nodes.dmp" "$ACC" > "$LINEAGE"
fi
exit 0
