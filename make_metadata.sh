#!/bin/sh
# make_metadata.sh — per-database metadata for set_targets.sh, same files as CuCLARK's script (make_metadata.sh:24-125):
#   ./make_metadata.sh <bacteria|viruses|human|custom> <database directory>
# <dir>/.<db>                     list of sequence files (custom: every <dir>/Custom/*.f*)
# <dir>/.<db>.fileToAccssnTaxID   file, accession, taxonomy ID          (exe/getAccssnTaxID)
# <dir>/.<db>.fileToTaxIDs        file, taxonomy ID, species..phylum IDs (exe/getfilesToTaxNodes)
# Taxonomy dumps are expected in <dir>/taxonomy (nodes.dmp, merged.dmp, nucl_accss) with the marker <dir>/.taxondata;
# when they or the sequences are missing the download scripts are called (they need network access).
HERE=$(dirname "$0")
if [ $# -lt 2 ] || [ -z "$2" ]; then
  echo "Usage: $0 <Database: bacteria, viruses, human or custom> <Directory path>"
  exit 0
fi
DB=$1
DBDR=$2
TAX="$DBDR/taxonomy"
if [ ! -d "$DBDR" ]; then
  echo "Selected directory not found. The program will create it."
  mkdir -m 775 "$DBDR"
fi
if [ ! -d "$DBDR" ]; then
  echo "Failed to find the directory (please check the name of directory $DBDR: Does it exist?). The program will abort."
  exit 1
fi
[ -d "$DBDR/Custom" ] || mkdir -m 775 "$DBDR/Custom"

if [ ! -d "$TAX" ]; then
  echo "Taxonomy data missing. The program will download data to $TAX."
  mkdir -m 775 "$TAX"
  "$HERE/download_taxondata.sh" "$TAX"
fi
if [ ! -f "$DBDR/.taxondata" ]; then
  echo "Failed to find taxonomy files. The program will try to download them..."
  "$HERE/download_taxondata.sh" "$TAX"
  if [ ! -f "$DBDR/.taxondata" ]; then
    echo "Failed to find taxonomy files."
    echo "The program must abort."
    exit 1
  fi
fi

if [ ! -s "$DBDR/.$DB" ]; then
  if [ "$DB" != "custom" ]; then
    echo "Sequences for $DB not found. The program will download them."
    "$HERE/download_data.sh" "$DBDR" "$DB"
  else
    find "$DBDR/Custom/" -name '*.f*' > "$DBDR/.$DB"
    if [ ! -s "$DBDR/.$DB" ]; then
      echo "The database directory 'Custom' is empty."
      echo "If you want CLARK to use a customized database then please do the following directions: "
      echo "1) Move your sequences in fasta format with the Accession number to $DBDR/Custom/"
      echo "2) Run again this command with the option 'custom' "
      exit 1
    fi
  fi
fi
if [ ! -x "$HERE/exe/getfilesToTaxNodes" ] || [ ! -x "$HERE/exe/getAccssnTaxID" ]; then
  echo "Something wrong occurred (source code may be missing or unusable. Did the installation finish properly?). The program must abort."
  exit 1
fi
if [ ! -s "$DBDR/.$DB" ]; then
  echo "Failed to find the downloaded $DB sequences."
  echo "The program must abort."
  exit 1
fi

if [ "$DB" = "human" ]; then   # one fixed lineage: Homo sapiens .. Chordata
  if [ ! -s "$DBDR/.$DB.fileToTaxIDs" ]; then
    while read -r file; do echo "$file X 9606 9605 9604 9443 40674 7711"; done < "$DBDR/.$DB" > "$DBDR/.$DB.fileToTaxIDs"
  fi
  exit 0
fi
if [ ! -s "$DBDR/.$DB.fileToAccssnTaxID" ]; then
  echo "Re-building $DB.fileToAccssnTaxID"
  "$HERE/exe/getAccssnTaxID" "$DBDR/.$DB" "$TAX/nucl_accss" "$TAX/merged.dmp" > "$DBDR/.$DB.fileToAccssnTaxID"
fi
if [ ! -s "$DBDR/.$DB.fileToTaxIDs" ]; then
  echo "$DB: Retrieving taxonomy nodes for each sequence based on taxon ID..."
  "$HERE/exe/getfilesToTaxNodes" "$TAX/nodes.dmp" "$DBDR/.$DB.fileToAccssnTaxID" > "$DBDR/.$DB.fileToTaxIDs"
fi
exit 0
