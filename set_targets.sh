#!/bin/sh
# set_targets.sh — defines the classification targets of a database directory.  Same arguments and same files as
# CuCLARK's script of this name (set_targets.sh:32-126), so either can prepare a directory for the other:
#   ./set_targets.sh <database directory> <bacteria|viruses|human|custom>+ [--species|--genus|--family|--order|--class|--phylum]
# Per selected database: metadata via make_metadata.sh, then "<file>\t<taxid at the rank>" lines appended to
# <dir>/targets.txt by exe/getTargetsDef.  Also written: ./.settings ("-T <dir>/targets.txt", "-D <dir>/<sub>/") for
# classify_metagenome.sh, ./.DBDirectory for the housekeeping scripts, <dir>/files_excluded.txt (sequences without a
# taxonomy ID) and the directory <dir>/<db1>_..._<rank number>_canonical/ that will hold the k-mer database.
HERE=$(dirname "$0")
if [ $# -lt 2 ] || [ -z "$1" ]; then
  echo "Usage: $0 <Directory path> <Databases: bacteria, viruses, human or custom>+ <taxonomy rank: --phylum, --class, --order, --family, --genus or --species (default)>"
  exit 0
fi
DBDR=$1
shift

# rank: the first rank option wins; anything else starting with -- is an error
RANK=0
for arg in "$@"; do
  case "$arg" in
    --species) RANK=0; break ;;
    --genus) RANK=1; break ;;
    --family) RANK=2; break ;;
    --order) RANK=3; break ;;
    --class) RANK=4; break ;;
    --phylum) RANK=5; break ;;
    --*) echo "Failed to recognize this parameter: $arg"; exit 1 ;;
  esac
done

if [ ! -d "$DBDR" ]; then
  echo "Database directory $DBDR does not exist yet: creating it."
  mkdir -m 775 "$DBDR" || { echo "Cannot create $DBDR (check the path)."; exit 1; }
fi
echo "$DBDR" > .DBDirectory

TARGETS="$DBDR/targets.txt"
EXCLUDED="$DBDR/files_excluded.txt"
for stale in "$TARGETS" "$EXCLUDED" "$DBDR/.tmp" .settings files_excluded.txt; do
  [ -f "$stale" ] && unlink "$stale"
done
: > "$TARGETS"
SUBDB=""
for db in "$@"; do
  case "$db" in --*) continue ;; esac
  printf "Collecting metadata of %s... " "$db"
  "$HERE/make_metadata.sh" "$db" "$DBDR" || exit 1
  [ -s "$DBDR/.$db" ] && [ -f "$DBDR/.taxondata" ] || exit 1
  echo "done."
  if [ -s "$DBDR/.$db.fileToTaxIDs" ]; then
    # getTargetsDef lists the files it leaves out in ./files_excluded.txt (its exit code is their number)
    "$HERE/exe/getTargetsDef" "$DBDR/.$db.fileToTaxIDs" $RANK >> "$TARGETS"
    SUBDB="${SUBDB}${db}_"
    if [ -f files_excluded.txt ]; then
      cat files_excluded.txt >> "$DBDR/.tmp"
      unlink files_excluded.txt
    fi
  fi
done
SUBDB="${SUBDB}${RANK}_canonical"
if [ ! -d "$DBDR/$SUBDB" ]; then
  echo "Creating directory to store discriminative k-mers: $DBDR/$SUBDB"
  mkdir -m 775 "$DBDR/$SUBDB"
fi
printf -- "-T %s\n-D %s/\n" "$TARGETS" "$DBDR/$SUBDB" > .settings
if [ -s "$DBDR/.tmp" ]; then
  mv "$DBDR/.tmp" "$EXCLUDED"
elif [ -f "$DBDR/.tmp" ]; then
  unlink "$DBDR/.tmp"
fi
